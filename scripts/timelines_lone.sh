cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, program args...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$name -o t -- python3 "$@" > gpurun_out/r03_f_${name}_trace.log 2>&1 || return 1
  f=$(find gpurun_out/prof_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name: $*" >> gpurun_out/r03_f_timelines.log
  python3 scripts/frame_timeline.py $f >> gpurun_out/r03_f_timelines.log 2>&1
  rm -rf gpurun_out/prof_$name
}
rm -f gpurun_out/r03_f_timelines.log
run single scripts/frames_only.py 40 && run lone_onechain scripts/frames_lone_slab.py 40 SPLIT_ROWS=0 && run lone_eager scripts/frames_lone_slab.py 40 USE_GRAPH=0 && run lone_eager_defer0 scripts/frames_lone_slab.py 40 USE_GRAPH=0 SLAB_DEFER=0
cat gpurun_out/r03_f_timelines.log
