"""Probe (MI355X, ROCm 7.x + PyTorch 2.10): ordering of hipGraph replays with eager launches and staging copies of the same stream.

x goes through  x = 2 x  (eager launch)  and a replayed chain of NODES launches  x = x + d  alternately, R rounds; d is a device scalar
staged from pageable host memory before every replay (like the engine's per-step inputs).  A correct execution ends at a known value in
every element.  Variants: eager launches / graph nodes from this library (handle 0 through the C ABI) or from torch; default or explicit
stream; stream-level host synchronization (.item() of a small tensor) between the phases or none.
usage: graph_order_probe.py"""
import os, sys
sys.path.insert(0, '.')
import torch
from video_prediction_amd import kernels as K

dev = torch.device('cuda:0')
n = 8 << 20
NODES = 200


def run(eager, nodes, explicit, item_sync, rounds=10):
    x = torch.zeros(n, device=dev)
    d = torch.zeros(n, device=dev)
    small = torch.zeros(4, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin()
        for _ in range(NODES):
            if nodes == 'lib':
                K.axpby(1.0, x, 1.0, d, x)
            else:
                x.add_(d)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    x.zero_()
    torch.cuda.synchronize()
    main = torch.cuda.Stream() if explicit else torch.cuda.current_stream()
    want = 0.0
    with torch.cuda.stream(main):
        for r in range(rounds):
            if eager == 'lib':
                K.axpby(0.5, x, 0.0, None, x)
            else:
                x.mul_(0.5)
            want = want * 0.5
            if item_sync:
                small.sum().item()
            inc = float(r % 3 + 1)
            d.copy_(torch.full((n,), inc))                 # pageable host -> device, like the engine's staged inputs
            g.replay()
            want = want + NODES * inc
            if item_sync:
                small.sum().item()
        if eager == 'lib':
            K.axpby(0.5, x, 0.0, None, x)
        else:
            x.mul_(0.5)
        want *= 0.5
    torch.cuda.synchronize()
    bad = int(((x - want).abs() > 1e-3 * abs(want)).sum())
    return bad, float(x.min()), float(x.max()), want


for eager in ('torch', 'lib'):
    for nodes in ('torch', 'lib'):
        for explicit in (False, True):
            for item_sync in (False, True):
                bad, lo, hi, want = run(eager, nodes, explicit, item_sync)
                print('eager=%-5s nodes=%-5s stream=%-8s host-sync=%-5s wrong elements %9d of %d  (min %.3f max %.3f want %.3f)' % (
                    eager, nodes, 'explicit' if explicit else 'default', 'item' if item_sync else 'none', bad, n, lo, hi, want))
