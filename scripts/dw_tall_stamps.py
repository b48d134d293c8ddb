"""s_memtime stamps of one workgroup (middle of the grid) of dwconv7_ln_tall_kernel: where a half-image tile's time goes.
Investigation build:  GP_EXTRA_HIPCC_FLAGS=-DGP_DW_STAMPS GP_BUILD_TAG=dwstamps python -m givepose_amd.build
                      GP_LIB_PATH=$PWD/givepose_amd/libgivepose_hip_dwstamps.so python scripts/dw_tall_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops, _lib
lib = _lib.load()
lib.gp_dwt_stamps_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
for (B, C, H, act) in ((128, 512, 16, 110), (128, 512, 16, 111), (64, 512, 16, 110), (128, 256, 16, 110)):   # 111: no MFMAs
    NS = C // 64
    x = torch.randn(B, H, 16, C, device="cuda").half(); w = (torch.randn(49, C, device="cuda") / 7).half()
    b, lw, lb = (torch.randn(C, device="cuda") for _ in range(3))
    y = torch.empty_like(x)
    for _ in range(5): ops.dwconv_ln(x, w, b, lw, lb, y, 7, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.dwconv_ln(x, w, b, lw, lb, y, 7, act=act)
    e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert lib.gp_dwt_stamps_read(buf) == 0
    print(f"C={C} {H}x16 B={B} act={act}: kernel {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; stamps of one workgroup, cycles since wave start")
    for wv in (0, 3, 7):
        t = lambda k: buf[wv * 64 + k] - buf[wv * 64 + 0]
        line = f"   wave {wv}: DMA plan + params + slabs 0-2 issued {t(1)} | zero fill + lane setup {t(2)} | slab 0 landed {t(4)} | first barrier {t(5)}"
        for s in range(NS):
            line += f"\n      slab {s}: start {t(6 + 4 * s)}"
            if s + 1 < NS:
                line += f" at the barrier (6 of 7 column shifts issued) {t(8 + 4 * s)} past it + DMA of slab s+3 issued {t(9 + 4 * s)}"
            line += f" conv done {t(7 + 4 * s)}  (slab {t(7 + 4 * s) - t(6 + 4 * s)})"
        line += f"\n      LN partials written {t(40)} barrier {t(41)} half A staged {t(42)} barrier {t(43)} half B staged (A's stores issued) {t(44)} barrier {t(45)} stores issued {t(46)} drained {t(47)}"
        print(line, flush=True)
