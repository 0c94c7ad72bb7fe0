"""Which kind of command keeps the HSA runtime's event thread busy?  Each mode runs ~3 s on one stream (or two) and prints the CPU seconds of the
process' threads other than this one.  usage: python tools/r06_rt_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
x = torch.zeros(1 << 22, device=dev); y = torch.zeros(1 << 22, device=dev)
hp = torch.zeros(1 << 22).pin_memory(); hq = torch.zeros(1 << 22).pin_memory()
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
me = None
def others(t0, t1):
    import threading
    tid = threading.get_native_id()
    return sorted(((round(t1[t][1] - t0.get(t, ("", 0.0))[1], 2), t1[t][0]) for t in t1 if t != tid), reverse=True)[:3]
def run(name, body, secs=3.0):
    torch.cuda.synchronize()
    t0 = bench.thread_cpu_seconds(); w0 = time.time(); n = 0
    while time.time() - w0 < secs:
        body(); n += 1
    torch.cuda.synchronize()
    t1 = bench.thread_cpu_seconds()
    print("%-58s %6d iterations in %.1f s; busiest other threads (CPU s): %s" % (name, n, time.time() - w0, others(t0, t1)), flush=True)
def poll(stream):
    while not stream.query(): time.sleep(0.0001)
def k_only():
    with torch.cuda.stream(s1):
        for _ in range(20): y.add_(x)
    poll(s1)
def k_sync():
    with torch.cuda.stream(s1):
        for _ in range(20): y.add_(x)
    s1.synchronize()
def k_h2d():
    with torch.cuda.stream(s1):
        x.copy_(hp, non_blocking=True)
        for _ in range(20): y.add_(x)
    poll(s1)
def k_d2h():
    with torch.cuda.stream(s1):
        for _ in range(20): y.add_(x)
        hq.copy_(y, non_blocking=True)
    poll(s1)
def k_events():
    with torch.cuda.stream(s2):
        x.add_(1.0); e = torch.cuda.Event(); e.record(s2)
    with torch.cuda.stream(s1):
        s1.wait_event(e)
        for _ in range(20): y.add_(x)
    poll(s1)
def k_event_query():
    with torch.cuda.stream(s1):
        for _ in range(20): y.add_(x)
        e = torch.cuda.Event(); e.record(s1)
    while not e.query(): time.sleep(0.0001)
def idle():
    time.sleep(0.01)
A = torch.zeros(4096, 4096, device=dev); B = torch.zeros(4096, 4096, device=dev)
def long_kernel(stream, n=6):
    with torch.cuda.stream(stream):
        for _ in range(n): torch.mm(A, B)           # a few ms each in fp32
def long_eventpoll():
    long_kernel(s1)
    with torch.cuda.stream(s1): e = torch.cuda.Event(); e.record(s1)
    while not e.query(): time.sleep(0.0001)
def long_cross():
    long_kernel(s2)
    with torch.cuda.stream(s2): e = torch.cuda.Event(); e.record(s2)
    with torch.cuda.stream(s1):
        s1.wait_event(e); y.add_(x); e2 = torch.cuda.Event(); e2.record(s1)
    while not e2.query(): time.sleep(0.0001)
def long_h2d_other_stream():
    long_kernel(s2)
    with torch.cuda.stream(s1):
        x.copy_(hp, non_blocking=True); e2 = torch.cuda.Event(); e2.record(s1)
    with torch.cuda.stream(s2): e = torch.cuda.Event(); e.record(s2)
    while not (e.query() and e2.query()): time.sleep(0.0001)
def long_h2d_then_wait():
    with torch.cuda.stream(s1):
        x.copy_(hp, non_blocking=True); e1 = torch.cuda.Event(); e1.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(e1)
    long_kernel(s2)
    with torch.cuda.stream(s2): e = torch.cuda.Event(); e.record(s2)
    while not e.query(): time.sleep(0.0001)
run("idle (sleep)", idle)
run("6 long kernels + event, poll hipEventQuery", long_eventpoll)
run("6 long kernels on stream 2, stream 1 waits for their event", long_cross)
run("6 long kernels on stream 2, H2D copy on stream 1 meanwhile", long_h2d_other_stream)
run("H2D on stream 1, stream 2 waits for it, 6 long kernels", long_h2d_then_wait)
run("20 kernels, poll hipStreamQuery + usleep(100)", k_only)
run("20 kernels, hipStreamSynchronize", k_sync)
run("pinned H2D copy + 20 kernels, poll", k_h2d)
run("20 kernels + D2H copy to pinned, poll", k_d2h)
run("event recorded on stream 2, waited for on stream 1, poll", k_events)
run("20 kernels + event, poll hipEventQuery + usleep(100)", k_event_query)
