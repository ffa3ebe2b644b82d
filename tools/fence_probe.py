import time, torch
torch.cuda.init()
x = torch.zeros(1, device='cuda')
for _ in range(5):
    torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('synchronize() on an idle device: median %.1f us' % (sorted(ts)[100] * 1e6))
ev = torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); ev.record(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
print('event record (timing): median %.1f us' % (sorted(ts)[100] * 1e6))
ev2 = torch.cuda.Event()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); ev2.record(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
print('event record (no timing): median %.1f us' % (sorted(ts)[100] * 1e6))
ts = []
for _ in range(200):
    ev2.record(); t0 = time.perf_counter()
    while not ev2.query(): pass
    ts.append(time.perf_counter() - t0)
print('record -> query true on an idle device: median %.1f us' % (sorted(ts)[100] * 1e6))
# a tiny kernel: launch -> done latency
ts = []
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); x.add_(1); ev2.record()
    while not ev2.query(): pass
    ts.append(time.perf_counter() - t0)
print('launch one tiny kernel + event -> seen done: median %.1f us' % (sorted(ts)[100] * 1e6))
