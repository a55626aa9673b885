for cfg in "lstm_h0 0x712" "lstm_h1 0x711" "lstm_h2 0x311"; do set -- $cfg
for a in 0 4 8 32 44 2 1 3; do
  SHAPE=$1 MODE=fprop TILE=$2 SAVP_ABLATE=$a ITERS=20 python tests/micro_one.py 2>&1 | tail -1
done; done
