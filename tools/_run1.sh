cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_lat -o stats -- python $GRAFT_REPO_ROOT/tools/gpu_small_loop.py 250 200 50 > $GRAFT_REPO_ROOT/gpurun_out/lat_kitti.txt 2>&1
DB=$(find /tmp/prof_lat -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB > $GRAFT_REPO_ROOT/gpurun_out/lat_kitti_stats.md 2>&1
tail -5 $GRAFT_REPO_ROOT/gpurun_out/lat_kitti.txt; head -30 $GRAFT_REPO_ROOT/gpurun_out/lat_kitti_stats.md | cut -c1-150
