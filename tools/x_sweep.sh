for cfg in "nowin=0" "nowin=1"; do
  for c in m256 c2 c3 m256b8; do echo "== $c $cfg"; timeout -k 10 100 python3 tools/run_steps.py $c 400 $cfg stage; done
  echo "== c4 $cfg"; timeout -k 10 100 python3 tools/run_steps.py c4 60 $cfg stage
done
