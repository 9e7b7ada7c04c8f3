# same-box A/B of several builds of libdl4vc_dan.so on one bench command:  tools/lib_ab.sh "<bench args>" <rounds> <build> [<build> ...]
#   <build> = tree (the tree's library) or NAME for tools/ab/libdl4vc_dan_NAME.so, e.g. built from HEAD's sources with other flags:
#     git archive HEAD dl4vc_amd/csrc include | tar -x -C /tmp/b && make -C /tmp/b/dl4vc_amd/csrc libdl4vc_dan.so CXXFLAGS='...' && cp ... tools/ab/
set -e
ARGS="${1:---precision 1 --sites 32768}"
N=${2:-2}
TAG=$(echo "$ARGS" | tr -c "a-zA-Z0-9" "_")
shift 2 || true
BUILDS="${@:-base tree}"
for i in $(seq 1 $N); do
  for v in $BUILDS; do
    if [ $v = tree ]; then unset DL4VC_DAN_LIB; else export DL4VC_DAN_LIB=$PWD/tools/ab/libdl4vc_dan_$v.so; fi
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path $ARGS > gpurun_out/ab_${TAG}_${v}_$i.json 2> gpurun_out/ab_${TAG}_${v}_$i.err || { tail -5 gpurun_out/ab_${TAG}_${v}_$i.err; exit 1; }
    python - $v $i $TAG <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_%s_%s_%s.json'%(sys.argv[3],sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1]); r=d.get('roofline',{})
p=d.get('parity',{})
print('%-8s %s %10.1f ms/step %9.2f seg launch ms %s parity %s %s %s' % (sys.argv[1],sys.argv[2],d['value'],d['ms_per_step'],r.get('avg_launch_ms'),p.get('ok'),p.get('tiled_identical'),p.get('max_abs_vt_prob')))
PY
  done
done
