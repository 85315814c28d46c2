"""CPU experiment for DESIGN §7: how close does a 3-way bf16 split of both fp32 operands (six bf16 products, fp32
accumulation - what six v_mfma_f32_16x16x32_bf16 per k-step would compute) come to a plain fp32 product?  Reference is fp64."""
import torch

torch.manual_seed(0)
M, K, N = 256, 12621, 256                       # K of the context Conv1D implicit GEMM


def split3(x):
    a1 = x.to(torch.bfloat16)
    r = x - a1.float()
    a2 = r.to(torch.bfloat16)
    a3 = (r - a2.float()).to(torch.bfloat16)
    return a1.float(), a2.float(), a3.float()


def mm32(a, b, chunk=32):
    """fp32 accumulation over k in MFMA-sized chunks (each chunk product is exact enough in fp32 for bf16 inputs)."""
    acc = torch.zeros(a.shape[0], b.shape[1])
    for k in range(0, a.shape[1], chunk):
        acc += a[:, k:k + chunk] @ b[k:k + chunk]
    return acc


a = torch.randn(M, K)
b = torch.randn(K, N) * 0.02
ref = a.double() @ b.double()
scale = ref.abs().mean()
f32 = mm32(a, b)
A, B = split3(a), split3(b)
terms = {3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)],
         9: [(i, j) for i in range(3) for j in range(3)]}
print('fp32        max err / mean|ref| = %.3e' % ((f32.double() - ref).abs().max() / scale))
for n, tt in terms.items():
    acc = torch.zeros(M, N)
    for i, j in sorted(tt, key=lambda t: -(t[0] + t[1])):       # small terms first
        acc += mm32(A[i], B[j])
    print('bf16x%d      max err / mean|ref| = %.3e' % (n, (acc.double() - ref).abs().max() / scale))
