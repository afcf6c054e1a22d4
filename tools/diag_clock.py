"""Diagnostic: engine clock (rocm-smi) while the dominant kernels run back to back - context for the roofline
fraction (the 2.5 PFLOP/s peak assumes the maximum clock; cdna_hip_programming.md rule 28)."""
import sys, os, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        for line in out.splitlines():
            if "sclk" in line.lower():
                return line.strip()
    except Exception as e:
        return f"rocm-smi failed: {e}"
    return "no sclk line"

print("idle      :", sclk())
q = torch.randn(12544, 1024, device="cuda").bfloat16(); k = torch.randn(6272, 1024, device="cuda").bfloat16(); v = torch.randn(6272, 1024, device="cuda").bfloat16()
a = torch.randn(12544, 4096, device="cuda").bfloat16(); w = torch.randn(1024, 4096, device="cuda").bfloat16(); b = torch.zeros(1024, device="cuda")
for name, fn in (("attention", lambda: ops.attention(q, k, v, 8, want_lse=True)), ("gemm K=4096", lambda: ops.linear(a, w, b))):
    stop = [False]
    def load():
        while not stop[0]:
            for _ in range(200): fn()
            torch.cuda.synchronize()
    t = threading.Thread(target=load); t.start()
    time.sleep(2.0)
    samples = [sclk() for _ in range(3)]
    stop[0] = True; t.join()
    print(f"{name:12s}:", " | ".join(samples))
