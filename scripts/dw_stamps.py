"""s_memtime stamps of one workgroup (the one in the middle of the grid) of dwconv7_ln_mfma_kernel: where a tile's time goes.
Investigation build:  GP_EXTRA_HIPCC_FLAGS=-DGP_DW_STAMPS GP_BUILD_TAG=dwstamps python -m givepose_amd.build
                      GP_LIB_PATH=$PWD/givepose_amd/libgivepose_hip_dwstamps.so python scripts/dw_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops, _lib
lib = _lib.load()
names = ["start", "DMA issued (+ index math)", "lane setup done", "vmcnt(0): own DMA landed", "barrier", "conv of all slabs done", "bias + LN partials written",
         "LN barrier", "statistics + staging written", "staging barrier", "stores issued", "stores drained"]
for (B, C, H) in ((128, 128, 64), (128, 256, 32), (128, 512, 16), (64, 512, 16)):
    x = torch.randn(B, H, H, C, device="cuda").half(); w = torch.randn(49, C, device="cuda").half()
    b, lw, lb = (torch.randn(C, device="cuda") for _ in range(3))
    y = torch.empty_like(x)
    for _ in range(5): ops.dwconv_ln(x, w, b, lw, lb, y, 7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.dwconv_ln(x, w, b, lw, lb, y, 7)
    e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    lib.gp_dw_stamps_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert lib.gp_dw_stamps_read(buf) == 0
    print(f"C={C} {H}x{H} B={B}: kernel {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; stamps of one workgroup (cycles since its start; waves 0 / 3 / 4 / 7)")
    for wv in (0, 3, 4, 7):
        t = [buf[wv * 16 + k] for k in range(12)]
        print("   wave %d: " % wv + "  ".join(f"{names[k].split(':')[0][:18]}={t[k] - t[0]}" for k in range(1, 12)))
    t = [buf[k] for k in range(12)]
    print("   wave 0 deltas: " + " | ".join(f"{names[k]}: {t[k] - t[k - 1]}" for k in range(1, 12)))
