set -e
for rm in 16 32 48; do
  echo "== YART_WIDE_REFILL_MIN=$rm"
  YART_WIDE_REFILL_MIN=$rm timeout -k 10 200 python -u tools/exp_pool.py --steps 3 --configs "11,10,512,32,32,24;11,10,512,24,32,24;11,10,512,16,32,24;11,10,512,1,32,24;5,10,256,24,32,24"
done
for s in 0.002 0.01; do
  echo "== s=$s"
  timeout -k 10 200 python -u tools/exp_pool.py --s $s --steps 3 --configs "11,10,512,32,32,24;11,10,512,16,32,24"
done
