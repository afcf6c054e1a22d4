"""Diagnostic (not a test): one training step of the memory path at the bench shape - forward under autograd +
backward, 64 frames, M=64, D=1024, bf16 - wall time and per-kernel-kind breakdown (HIP events).
usage: python tools/bench_train.py [steps]"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi
from memory_augmented_vlm_amd.model.memory_module.MemoryController import Config, TransformerProjector

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
capi.lib().mavlm_set_attention_bwd_fused(int(os.environ.get("BWD_FUSED", "0")))
M = int(os.environ.get("MEM_TOKENS", "64"))
HID = int(os.environ.get("HIDDEN", "1024"))
c = Config(); c.mm_hidden_size = HID; c.mm_intermediate_size = 4 * HID; c.mm_num_attention_heads = 8
c.num_memory_tokens = M; c.patch_size = 196; c.depth = 2; c.mm_dtype = torch.float32
torch.manual_seed(0)
rm = TransformerProjector(c).cuda().to(torch.bfloat16).train()
x = (torch.randn(64, 196, HID, device="cuda") * 0.5).bfloat16()

def step():
    rm.zero_grad(set_to_none=True)
    rm.memory_cache = []
    for i in range(2):
        cache, _ = rm(x[32 * i:32 * i + 32])
    loss = sum(cm.float().square().mean() for cm in cache)
    loss.backward()
    return loss

def fwd_only():
    with torch.no_grad():
        rm.memory_cache = []
        for i in range(2):
            rm(x[32 * i:32 * i + 32])

for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
for _ in range(2): fwd_only()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): fwd_only()
torch.cuda.synchronize(); df = (time.perf_counter() - t0) / steps
print(f"train step (fwd+bwd) {dt*1e3:.2f} ms   inference fwd {df*1e3:.2f} ms   ratio {dt/df:.2f}   peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
lib = capi.lib(); nk = len(capi.KERNEL_KINDS)
ms = (ctypes.c_double * nk)(); ln = (ctypes.c_int64 * nk)(); fl = (ctypes.c_double * nk)(); by = (ctypes.c_double * nk)()
lib.mavlm_prof_enable(1)
for _ in range(steps): step()
torch.cuda.synchronize()
capi.check(lib.mavlm_prof_read(ms, ln, fl, by, nk), "prof"); lib.mavlm_prof_enable(0)
tot = sum(ms)
for i, name in enumerate(capi.KERNEL_KINDS):
    if ln[i]:
        print(f"{name:18s} launches/step {ln[i]/steps:6.1f}  ms/step {ms[i]/steps:8.3f}  ({100*ms[i]/tot:4.1f}%)  avg {ms[i]/ln[i]*1e3:8.1f} us"
              + (f"  {fl[i]/(ms[i]*1e-3)/1e12:7.1f} TF" if fl[i] else f"  {by[i]/(ms[i]*1e-3)/1e9:7.1f} GB/s"))
print(f"sum of HIP kernels {tot/steps:.2f} ms/step (torch glue kernels not included)")
