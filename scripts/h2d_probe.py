"""Which host->device upload calls block the host while the stream is busy? (MI355X / ROCm 7.2 / torch 2.10)"""
import time, torch, numpy as np
dev = torch.device('cuda:0')
a = torch.randn(8192, 8192, device=dev)
def busy():
    for _ in range(6):
        a @ a          # ~7 ms each
def t(name, fn):
    torch.cuda.synchronize(); busy()
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f'{name:60s} {dt * 1e3:8.3f} ms'); return r
x_np = np.random.rand(6, 16).astype(np.float32)
pin = torch.empty(6, 16).pin_memory()
pin.numpy()[...] = x_np
print('is_pinned', pin.is_pinned())
d = torch.empty(6, 16, device=dev)
big_pin = torch.empty(1 << 20).pin_memory()
big_d = torch.empty(1 << 20, device=dev)
t('baseline: enqueue nothing', lambda: None)
t('pageable .to(non_blocking)', lambda: torch.from_numpy(x_np).to(dev, non_blocking=True))
t('pinned .to(non_blocking) 384 B', lambda: pin.to(dev, non_blocking=True))
t('pinned copy_ into existing 384 B', lambda: d.copy_(pin, non_blocking=True))
t('pinned copy_ 4 MiB', lambda: big_d.copy_(big_pin, non_blocking=True))
ev = torch.cuda.Event()
t('event record', lambda: ev.record())
t('torch.empty on device', lambda: torch.empty(6, 16, device=dev))
t('kernel reading pinned host memory directly (add)', lambda: torch.add(d, 1.0, out=d))
# zero-copy: a device tensor aliasing pinned host memory is not expressible in torch; time a fill kernel as the stand-in
