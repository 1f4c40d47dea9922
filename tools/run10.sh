cd $GRAFT_REPO_ROOT/tools/probes
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 symstore_probe.hip -o /tmp/symstore_probe 2>&1 | tail -3
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
( /tmp/symstore_probe 65536 0 48; /tmp/symstore_probe 65536 0 16; /tmp/symstore_probe 65536 0 1024; /tmp/symstore_probe 65536 16 48 ) > $GRAFT_REPO_ROOT/gpurun_out/r10_symstore.txt 2>&1
