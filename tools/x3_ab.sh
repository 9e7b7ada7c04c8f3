# same-box A/B of two builds of libdl4vc_dan.so on a bench command: tools/x3_ab.sh "<bench args>" [rounds]
#   base = tools/ab/libdl4vc_dan_base.so (built from HEAD's sources: git archive HEAD dl4vc_amd/csrc include | tar -x -C /tmp/oldsrc; make), new = the tree's
set -e
ARGS="${1:---precision 1 --sites 32768}"
N=${2:-2}
for i in $(seq 1 $N); do
  for v in base new; do
    if [ $v = base ]; then export DL4VC_DAN_LIB=$PWD/tools/ab/libdl4vc_dan_base.so; else unset DL4VC_DAN_LIB; fi
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-skip-pass --no-host-path $ARGS > gpurun_out/ab_${v}_$i.json 2> gpurun_out/ab_${v}_$i.err || { tail -5 gpurun_out/ab_${v}_$i.err; exit 1; }
    python - $v $i <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1],sys.argv[2],d['value'],'ms/step',d['ms_per_step'],'seg launch ms',r['avg_launch_ms'],'parity',d['parity']['ok'],d['parity']['tiled_identical'],d['parity']['max_abs_vt_prob'],r['other_kernels_ms_per_step'],d['build']['source_hash'])
PY
  done
done
