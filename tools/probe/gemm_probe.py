"""What the vendor GEMM library (hipBLASLt / rocBLAS behind torch.matmul) does on the plain-GEMM layers of the path -- the OSA
concat 1x1 convolutions and the FPN laterals -- next to this repo's gemm1x1_pipe_kernel (tools/layer_report.py rows).  A probe
for DESIGN.md, not part of the product path."""
import json
import sys
import torch

SHAPES = [("s2.cat", 268800, 256, 768), ("s3.b0.cat", 67200, 512, 1056), ("s3.b1.cat", 67200, 512, 1312),
          ("s4.b0.cat", 16800, 768, 1472), ("s4.b1.cat", 16800, 768, 1728), ("s5.b0.cat", 4200, 1024, 1888),
          ("s5.b1.cat", 4200, 1024, 2144), ("fpn.lat3", 67200, 256, 512), ("fpn.lat4", 16800, 256, 768)]


def main():
    dev = torch.device("cuda", 0)
    out = []
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        fn = lambda: torch.relu_(torch.addmm(b, a, w.t()))       # bias + ReLU like the folded BN epilogue (two extra passes at worst)
        fn2 = lambda: torch.matmul(a, w.t())
        row = {"layer": name, "M": M, "N": N, "K": K}
        for tag, f in (("matmul", fn2), ("addmm_relu", fn)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    f()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            row[tag + "_us"] = round(us, 1)
            row[tag + "_tflops"] = round(2.0 * M * N * K / us / 1e6, 1)
        out.append(row)
        print(json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            for r in out:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
