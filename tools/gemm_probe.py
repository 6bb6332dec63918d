"""Micro-probe of the skinny head GEMMs of the PPO update on the GPU (which formulation the BLAS library runs well)."""
import torch, time
dev = torch.device('cuda')
R = 750 * 4096
x = torch.randn(R, 48, device=dev)
dy12 = torch.randn(R, 12, device=dev)
dy1 = torch.randn(R, 1, device=dev)


def t(name, f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) * 100, 3), 'ms')


def tall(a, b, chunks):
    K = a.shape[0]
    return torch.bmm(a.reshape(chunks, K // chunks, a.shape[1]).transpose(1, 2), b.reshape(chunks, K // chunks, b.shape[1])).sum(0)


def tall_t(a, b, chunks):
    K = a.shape[0]
    return torch.bmm(b.reshape(chunks, K // chunks, b.shape[1]).transpose(1, 2), a.reshape(chunks, K // chunks, a.shape[1])).sum(0).t()


for dy, nm in ((dy12, 'pi'), (dy1, 'vf')):
    for ch in (256, 1024, 4096):
        t('dW %s tall chunks=%d' % (nm, ch), lambda: tall(x, dy, ch))
        t('dW %s tall_t chunks=%d' % (nm, ch), lambda: tall_t(x, dy, ch))
    t('dW %s plain' % nm, lambda: x.t() @ dy)
t('dW vf as mv', lambda: torch.mv(x.t(), dy1[:, 0]))
t('dW vf mul-sum', lambda: (x * dy1).sum(0))
t('db pi sum', lambda: dy12.sum(0))
