set -e
for i in 1 2; do
for v in base tree; do
  if [ $v = tree ]; then unset DL4VC_DAN_LIB; else export DL4VC_DAN_LIB=$PWD/tools/ab/libdl4vc_dan_$v.so; fi
  python bench.py --mode train --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/tab_${v}_$i.json 2> gpurun_out/tab_${v}_$i.err || { tail -5 gpurun_out/tab_${v}_$i.err; exit 1; }
  python bench.py --mode train --train-batch 10 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/tab10_${v}_$i.json 2> gpurun_out/tab10_${v}_$i.err
  python - $v $i <<'PY'
import json,sys
for pre in ('tab','tab10'):
    d=json.loads(open('gpurun_out/%s_%s_%s.json'%(pre,sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
    print(pre,sys.argv[1],sys.argv[2],d['value'],'ms/step',d['ms_per_step'],'frac',d['roofline']['frac'])
PY
done
done
