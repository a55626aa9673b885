"""Probe (MI355X, ROCm 7.x + PyTorch 2.10): are hipMemsetAsync NODES of a captured graph ordered with the kernel nodes around them?

Chain of NODES x [ hipMemsetAsync(y, 0) ; y += x (atomic-free kernel) ; z += y ]: with every node in order z ends at NODES * x after a
replay (z cleared by an eager fill first).  Replayed R times with stream-level host synchronization only (.item()), like a training
loop that reads a loss; counts wrong elements per replay.  Sizes: y small (like a statistics workspace) or large.
usage: graph_memset_probe.py"""
import ctypes, sys
sys.path.insert(0, '.')
import torch
from video_prediction_amd import kernels as K, lib

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device('cuda:0')


def run(n, nodes, use_memset_node, rounds=30):
    x = torch.full((n,), 3.0, device=dev)
    y = torch.zeros(n, device=dev)
    z = torch.zeros(n, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin(capture_error_mode='thread_local')
        for _ in range(nodes):
            if use_memset_node:
                rc = hip.hipMemsetAsync(y.data_ptr(), 0, n * 4, lib.stream())
                assert rc == 0, rc
            else:
                y.zero_()
            K.axpby(1.0, x, 1.0, y, y)
            K.axpby(1.0, y, 1.0, z, z)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    worst = 0
    bad_rounds = 0
    for r in range(rounds):
        z.zero_()
        g.replay()
        bad = int((z != 3.0 * nodes).sum().item())
        worst = max(worst, bad)
        bad_rounds += bad > 0
    return bad_rounds, worst


def run_copy(n, nodes, rounds=30):
    """Chain of NODES x [ y.copy_(x) (hipMemcpyAsync D2D node) ; x += 1 ; z += y ]: z ends at sum_{k<NODES} (x0 + k)."""
    x = torch.zeros(n, device=dev)
    y = torch.zeros(n, device=dev)
    z = torch.zeros(n, device=dev)
    one = torch.ones(n, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin(capture_error_mode='thread_local')
        for _ in range(nodes):
            y.copy_(x)
            K.axpby(1.0, x, 1.0, one, x)
            K.axpby(1.0, y, 1.0, z, z)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    worst = bad_rounds = 0
    for r in range(rounds):
        z.zero_()
        x.zero_()
        g.replay()
        bad = int((z != nodes * (nodes - 1) / 2.0).sum().item())
        worst = max(worst, bad)
        bad_rounds += bad > 0
    return bad_rounds, worst


for n in (4096, 1 << 20):
    br, worst = run_copy(n, 50)
    print('n=%8d nodes=  50 copy =%-22s replays with wrong elements %2d of 30 (worst %d elements)' % (n, 'hipMemcpyAsync D2D node', br, worst))
for n in (4096, 1 << 20):
    for nodes in (50, 400):
        for mem in (False, True):
            br, worst = run(n, nodes, mem)
            print('n=%8d nodes=%4d clear=%-22s replays with wrong elements %2d of 30 (worst %d elements)' % (
                n, nodes, 'hipMemsetAsync node' if mem else 'fill kernel node', br, worst))
