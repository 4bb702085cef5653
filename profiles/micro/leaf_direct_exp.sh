# A/B of k_leaf_direct variants (MDB_LD_VAR: bit 0 = 16-byte emit scan, bit 1 = exact record reservation; MDB_LD_THREADS, MDB_LD_WGS)
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
for v in 0 1 2 3; do
  MDB_LD_VAR=$v $B > gpurun_out/r02_b5_D_var$v.json 2>>gpurun_out/r02_b5.err
  MDB_LD_VAR=$v $B --variant U > gpurun_out/r02_b5_U_var$v.json 2>>gpurun_out/r02_b5.err
done
MDB_LD_THREADS=256 $B > gpurun_out/r02_b5_D_t256.json 2>>gpurun_out/r02_b5.err
MDB_LD_THREADS=256 MDB_LD_WGS=4 $B > gpurun_out/r02_b5_D_t256w4.json 2>>gpurun_out/r02_b5.err
MDB_LD_WGS=3 $B > gpurun_out/r02_b5_D_w3.json 2>>gpurun_out/r02_b5.err
tail -2 gpurun_out/r02_b5.err
