"""lab: event record / wait NODES added by hand to the graphs under stream capture (hipStreamGetCaptureInfo_v2 + hipGraphAddEvent*Node +
hipStreamUpdateCaptureDependencies): does a wait node of one graph order behind the record node of another graph replayed before it?"""
import ctypes
import time
import torch

hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
hip.hipStreamGetCaptureInfo_v2.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(vp), ctypes.POINTER(ctypes.POINTER(vp)),
                                           ctypes.POINTER(ctypes.c_size_t)]
hip.hipGraphAddEventRecordNode.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_size_t, vp]
hip.hipGraphAddEventWaitNode.argtypes = [ctypes.POINTER(vp), vp, ctypes.POINTER(vp), ctypes.c_size_t, vp]
hip.hipStreamUpdateCaptureDependencies.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_size_t, ctypes.c_uint]
hip.hipGetErrorString.restype = ctypes.c_char_p


def add_node(stream, ev, record):
    st, cid, graph, deps, nd = ctypes.c_int(), ctypes.c_ulonglong(), vp(), ctypes.POINTER(vp)(), ctypes.c_size_t()
    rc = hip.hipStreamGetCaptureInfo_v2(stream.cuda_stream, ctypes.byref(st), ctypes.byref(cid), ctypes.byref(graph), ctypes.byref(deps), ctypes.byref(nd))
    assert rc == 0 and st.value == 1, (rc, st.value)
    node = vp()
    rc = (hip.hipGraphAddEventRecordNode if record else hip.hipGraphAddEventWaitNode)(ctypes.byref(node), graph, deps, nd.value, ev)
    assert rc == 0, (rc, hip.hipGetErrorString(rc))
    arr = (vp * 1)(node)
    rc = hip.hipStreamUpdateCaptureDependencies(stream.cuda_stream, arr, 1, 1)
    assert rc == 0, (rc, hip.hipGetErrorString(rc))


x = torch.zeros(1 << 24, device="cuda")
big = torch.zeros(1 << 28, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev = vp()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 2) == 0
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
import sys
BRANCH = len(sys.argv) > 1          # any argument: g1 also carries a parallel branch (a second captured stream), as the train step's graph does
s3 = torch.cuda.Stream()
other = torch.zeros(1 << 26, device="cuda")
with torch.cuda.graph(g1, stream=s1, capture_error_mode="thread_local"):
    if BRANCH:
        s3.wait_stream(s1)
        with torch.cuda.stream(s3):
            for _ in range(60):
                other.add_(1.0)
    for _ in range(20):
        big.add_(1.0)          # ~1 ms of work in front of the record
    x.add_(1.0)
    add_node(s1, ev, True)
    for _ in range(20):
        big.add_(1.0)          # ~1 ms of work behind the record: a concurrent g2 multiplies BEFORE the last add (x = 3)
    x.add_(1.0)
    if BRANCH:
        s1.wait_stream(s3)
with torch.cuda.graph(g2, stream=s2, capture_error_mode="thread_local"):
    add_node(s2, ev, False)
    x.mul_(2.0)
torch.cuda.synchronize()
for it in range(5):
    x.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        g1.replay()
    with torch.cuda.stream(s2):
        g2.replay()
    torch.cuda.synchronize()
    print(f"replay {it}: x = {x[0].item()}  (3 or 4 = the wait saw this replay's record; 2 = it passed on a stale record), {1e3 * (time.perf_counter() - t0):.2f} ms")
