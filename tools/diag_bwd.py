import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))   # gpu_util
import numpy as np, torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _ops as ops
from oracle import memory_path as O, backward as OB
from gpu_util import to_dev, to_np
r = O.bf16_round
for (R, S, H) in [(392, 6272, 2), (392, 784, 2), (392, 1024, 1)]:
    W = H * 128
    Q, K, V, dO = (r(O.hash_normal_like(s, i, sc)) for s, i, sc in (((R, W), 1, 1.0), ((S, W), 2, 1.0), ((S, W), 3, 1.0), ((R, W), 4, 0.5)))
    q, k, v, do = (to_dev(a) for a in (Q, K, V, dO))
    o, lse = ops.attention(q, k, v, H, want_lse=True)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, H)
    rq, rk, rv = OB.attention_bwd(Q, K, V, to_np(o), dO, to_np(lse), H, "bf16")
    print(R, S, H, "dq %.2e dk %.2e dv %.2e" % (O.rel_l2(to_np(dq), r(rq)), O.rel_l2(to_np(dk), r(rk)), O.rel_l2(to_np(dv), r(rv))))
for (M, N, K) in [(256, 256, 448), (256, 256, 6272), (1024, 256, 448), (256, 1024, 6272)]:
    A = np.floor(O.hash_uniform((M, K), 5, -2, 2.999)).astype(np.float32)
    B = np.floor(O.hash_uniform((N, K), 6, -2, 2.999)).astype(np.float32)
    got = to_np(ops.matmul_nt_splitk(to_dev(A), to_dev(B)))
    print(M, N, K, "splitk exact:", np.array_equal(got, r(A @ B.T)), "max|diff|", np.abs(got - r(A @ B.T)).max())
