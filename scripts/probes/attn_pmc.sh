set -u
root="$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ap /tmp/at
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/ap -o r -- python $root/scripts/probes/attn_all_one.py > /dev/null 2>&1
db=$(find /tmp/ap -name '*.db' | head -1)
python $root/scripts/pmc_mfma_util.py "$db" $root/gpurun_out/r06_attn_mfma_pmc.md "attention kernels on the cfg3 two-group scoring layout (28 / 4 heads x 128, P = 1402, 8 x 512 x 2), round-6 tree: MFMA pipe utilisation" attn_ > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/at -o r -- python $root/scripts/probes/attn_all_one.py > /dev/null 2>&1
db=$(find /tmp/at -name '*.db' | head -1)
python $root/scripts/rocprof_summary.py "$db" $root/gpurun_out/r06_attn_kernel_stats.md "attention kernels, cfg3 two-group layout, round-6 tree (scripts/probes/attn_all_one.py)" > /dev/null
cat $root/gpurun_out/r06_attn_mfma_pmc.md | tail -8; grep attn_ $root/gpurun_out/r06_attn_kernel_stats.md | head
