# kernel timelines of an inner slab (3072 x [rows=1536] of [world=4]) next to one context over the same rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, program args...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$name -o t -- python3 "$@" > gpurun_out/r03_t_${name}_trace.log 2>&1 || return 1
  f=$(find gpurun_out/prof_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name: $*" >> gpurun_out/r03_t_timelines.log
  python3 scripts/frame_timeline.py $f >> gpurun_out/r03_t_timelines.log 2>&1
  python3 scripts/kernel_means.py $f >> gpurun_out/r03_t_timelines.log 2>&1
  rm -rf gpurun_out/prof_$name
}
rm -f gpurun_out/r03_t_timelines.log
export FS_H=${1:-1536}
W=${2:-4}
run single$FS_H scripts/frames_only.py 60 && run inner scripts/frames_lone_slab.py 60 3072 $FS_H $W 1
cat gpurun_out/r03_t_timelines.log
