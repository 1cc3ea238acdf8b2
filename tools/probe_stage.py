"""L2 -> CU staging rates on this chip: LDS-DMA vs register loads vs both (ctx_probe_stage).  Prints one line per form."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from contexture_nerf_amd import _lib

def main():
    dev = torch.device("cuda:0")
    src = torch.randint(0, 255, (4 << 20,), dtype=torch.uint8, device=dev)
    sink = torch.zeros(16, dtype=torch.int32, device=dev)
    L = _lib.load()
    iters = 4000
    for shared in (0, 1):
        for waves in (4, 8, 16):
            for u in (4, 8):
                row = []
                for mode in (0, 1, 2):
                    ms = L.ctx_probe_stage(mode, waves, u, iters, shared, src.data_ptr(), sink.data_ptr(), None)
                    if ms <= 0:
                        raise SystemExit("probe failed: %s" % L.ctx_last_error().decode())
                    bytes_ = 256.0 * waves * u * 1024 * iters
                    row.append(bytes_ / (ms * 1e-3) / 1e12)
                print("shared_region=%d waves=%2d in_flight_KiB_per_wave=%d  lds_dma %.1f TB/s  regs %.1f TB/s  both %.1f TB/s  (per CU: %.0f / %.0f / %.0f GB/s)"
                      % (shared, waves, u, row[0], row[1], row[2], row[0] * 1e3 / 256, row[1] * 1e3 / 256, row[2] * 1e3 / 256), flush=True)

if __name__ == "__main__":
    main()
