import time, threading, numpy as np, torch
a = np.random.randn(64, 400, 601).astype(np.float32)
pin = torch.empty(a.size, dtype=torch.float32).pin_memory().view(a.shape)
d = torch.empty(a.shape, device='cuda')
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print('threads', torch.get_num_threads())
print('host->pinned copy_ ms', t(lambda: pin.copy_(torch.from_numpy(a))))
print('np.copyto into pinned ms', t(lambda: np.copyto(pin.numpy(), a)))
print('pinned->device ms', t(lambda: d.copy_(pin, non_blocking=True)))
print('pageable->device ms', t(lambda: d.copy_(torch.from_numpy(a))))
res = {}
def w():
    res['thr'] = t(lambda: pin.copy_(torch.from_numpy(a)))
th = threading.Thread(target=w); th.start(); th.join()
print('host->pinned in a thread ms', res['thr'])
