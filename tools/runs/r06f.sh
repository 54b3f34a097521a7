set -x
mkdir -p gpurun_out/r06f
cd $GRAFT_REPO_ROOT
for shape in "32 10" "8 60" "48 10" "32 15" "16 30"; do
  set -- $shape
  for m in "12=5" "12=0" "12=2" "12=0 --opt 11=-1" "12=2 --opt 11=-1"; do
    tag=$(echo "$1x$2_$m" | tr ' =-' '___')
    timeout 300 python bench.py --no-cpu-baseline --no-api --no-other-configs --batch $1 --clip-seconds $2 --opt $m > gpurun_out/r06f/$tag.json 2>> gpurun_out/r06f/err.txt
  done
done
timeout 300 python tools/api_timeline.py > gpurun_out/r06f/api_timeline.txt 2>&1
