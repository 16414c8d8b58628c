#!/usr/bin/env python3
"""What the vendor library does on the GEMM shapes of BASELINE configs[3] / configs[4] (f32 and bf16 torch.mm, i.e.
hipBLASLt / rocBLAS under PyTorch-ROCm): a yardstick for gemm_f32_kernel / gemm_bf16_kernel, not a dependency --
nothing in the product calls it.  One line per shape: us per product (HIP events over 50 launches) and TFLOP/s."""
import torch

SHAPES = [("configs[3] forward 1", 512, 2048, 4096, "nn"), ("configs[3] forward 2", 512, 2048, 2048, "nn"),
          ("configs[3] logits", 512, 1024, 2048, "nn"), ("configs[3] backward data 1", 512, 2048, 2048, "nt"),
          ("configs[3] gradient 0", 4096, 2048, 512, "tn"), ("configs[3] gradient 1", 2048, 2048, 512, "tn"),
          ("configs[4] forward", 256, 1024, 1024, "nn"), ("configs[4] backward data", 256, 1024, 1024, "nt"),
          ("configs[4] gradient", 1024, 1024, 256, "tn")]


def main():
    dev = torch.device("cuda:0")
    for dtype in (torch.float32, torch.bfloat16):
        for name, M, N, K, form in SHAPES:
            a = torch.randn((K, M) if form == "tn" else (M, K), device=dev, dtype=dtype)
            b = torch.randn((N, K) if form == "nt" else (K, N), device=dev, dtype=dtype)
            A = a.t() if form == "tn" else a
            B = b.t() if form == "nt" else b
            for _ in range(5):
                torch.mm(A, B)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                torch.mm(A, B)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            print("%-28s %-8s %4dx%4dx%4d  %8.2f us  %7.1f TFLOP/s" % (name, str(dtype).split(".")[1], M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
