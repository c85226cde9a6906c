# round 5: SQ counters of the persistent LDS-weights kernels against the one-tile kernels they replaced (GPU box, through gpurun)
cd "${GRAFT_REPO_ROOT:?}"
run() { echo "=== $1"; shift; env "$@" bash scripts/pmc_kernels.sh conv scripts/pmc_layer.py $L 2>&1 | grep -v "^ *SQ_\|amdgpu.ids" ; }
L="64 32 12 37 50 tr"
run "64->32 T @12x37x50: per-class conv3d_kernel<kTr>" MDF_CONVTR_ALL_MIN_VOXELS=-1
run "64->32 T @12x37x50: all classes per wave, n-tiles over four waves" MDF_CONVTR_CLS=0
run "64->32 T @12x37x50: one class per persistent block, slots in LDS (default)" MDF_X=0
L="32 16 24 74 100 tr"
run "32->16 T @24x74x100: all-classes one-tile kernel" MDF_CONVTR_WLDS=0
run "32->16 T @24x74x100: persistent LDS-weights form (default)" MDF_X=0
L="32 32 6 74 100 s1"
run "32->32 @6x74x100: split-K conv3d_kernel" MDF_CONV3D_WLDS=0
run "32->32 @6x74x100: conv3d_wlds_kernel (default)" MDF_X=0
L="8 16 8 592 800 s2"
run "8->16 s2 @8x592x800: conv3d_kernel" MDF_CONV3D_WLDS8=0
run "8->16 s2 @8x592x800: conv3d_wlds_kernel (default)" MDF_X=0
