for sh in lstm_h0 lstm_h1; do
for tile in 0x712 0x621; do
for a in 0 4 8 12 32 36 40 44 2 1 3; do
  SHAPE=$sh MODE=fprop TILE=$tile SAVP_ABLATE=$a ITERS=20 python tests/micro_one.py 2>&1 | tail -1
done; done; done
