"""Run one model-level check by expression, e.g.: python tests/tools/run_one_check.py "check_train_step(B=1,T=4,nz=8,steps=1,tag='g',conv_rnn='gru')" """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_model_checks as G  # noqa
from tests.gpu_model_checks import *  # noqa
res = eval(sys.argv[1])
bad = 0
for n, e, t in res:
    ok = e <= t
    bad += (not ok)
    print('%-6s %-60s err=%.3e tol=%.1e' % ('ok' if ok else 'FAIL', n, e, t))
print('FAILURES', bad)
