# development tool: kernel split of the affine NW path under rocprofv3, for ASM_RING_BYTES = 0 and 1
cd /tmp && export TMPDIR=/tmp
for rb in 0 1; do
  ASM_RING_BYTES=$rb rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nwaff_rb$rb -- python3 $GRAFT_REPO_ROOT/tools/bench_nw_affine.py C2 1e6 > $GRAFT_REPO_ROOT/gpurun_out/nwaff_rb$rb.txt 2>&1 || exit 1
done
