"""Phase timeline of ntt_mac_f64 from s_memtime stamps (debug build: bash tools/build_variant.sh stamps "-DHE_MAC_STAMPS=1").
usage (GPU box): HERING_LIB=lattigo_amd/variants/libhering_stamps.so python tools/mac_timeline.py"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import lattigo_amd as la
from lattigo_amd import _lib
from lattigo_amd.dist import ControlPlane
ctx = la.Context(0)
args = argparse.Namespace(replicate_keys="none")
W = bench.setup_c3(la, ctx, 0, 128, ControlPlane(), args)
for _ in range(3):
    W["step"]()
ctx.sync()
lib = _lib.load()
n = 2048 * 64
buf = (C.c_uint64 * n)()
rc = lib.he_debug_mac_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 64).astype(np.int64)
names = ["start", "r0pre", "r0post", "x0", "r1pre", "r1post", "x1", "r2pre", "r2post", "x2(bar)", "mac", "endbar"]
if "--dma" in sys.argv:  # the persistent LDS-DMA kernel: item HE_MAC_STAMP_ITEM of every workgroup
    names = ["start", "dma-wait+read", "tw2+dma issue", "round0", "bar+x1", "round1", "x2", "k0+round2+k1", "x3(bar)", "mac0", "mac1", "-"]
print("rc", rc, "nonzero rows", int((a[:, 0] > 0).sum()))
a = a[a[:, 0] > 0]
if "--dma" in sys.argv:  # the persistent kernel stamps a digit at: start, words read, transform + key rows done, MAC 0 done, MAC 1 done
    for dg in range(4):
        st = a[:, 1 + 12 * dg: 1 + 12 * dg + 12][:, [0, 1, 8, 9, 10]]
        ok = st[(st > 0).all(axis=1)]
        if len(ok):
            seg = np.diff(ok, axis=1)
            print("digit", dg, "waves", len(ok), dict(zip(["dma-wait+read", "transform+keys (own digit: keys only)", "mac0", "mac1"], (int(np.median(seg[:, i])) for i in range(4)))),
                  "sum", int(np.median(seg.sum(axis=1))))
for dg in range(4 if "--dma" not in sys.argv else 0):
    st = a[:, 1 + 12 * dg: 1 + 12 * dg + 12]
    full = st[(st[:, :11] > 0).all(axis=1)][:, :11] if "--dma" in sys.argv else st[(st > 0).all(axis=1)]
    dma = "--dma" in sys.argv
    own = st[(st[:, 2 if dma else 1] == 0) & (st[:, 0] > 0)]
    if len(full):
        seg = np.diff(full, axis=1)
        print("digit", dg, "waves", len(full), {names[i + 1]: int(np.median(seg[:, i])) for i in range(seg.shape[1])}, "sum", int(np.median(seg.sum(axis=1))))
    if len(own) and dma:
        print("digit", dg, "own waves", len(own), {"dma-wait+read": int(np.median(own[:, 1] - own[:, 0])), "dma+key issue": int(np.median(own[:, 8] - own[:, 1])),
              "mac0": int(np.median(own[:, 9] - own[:, 8])), "mac1": int(np.median(own[:, 10] - own[:, 9]))})
    elif len(own):
        print("digit", dg, "own waves", len(own), "start->mac", int(np.median(own[:, 10] - own[:, 0])), "mac->endbar", int(np.median(own[:, 11] - own[:, 10])))
if "--dma" in sys.argv and (a[:, 48] > 0).any():  # fused ModDown epilogue: the two extension phases
    e = a[(a[:, 48:56] > 0).all(axis=1)]
    for c in range(2):
        b = 48 + 4 * c
        print("ext", c, "waves", len(e), {"dma-wait+read": int(np.median(e[:, b + 1] - e[:, b])), "transform": int(np.median(e[:, b + 2] - e[:, b + 1])),
              "epilogue": int(np.median(e[:, b + 3] - e[:, b + 2]))})
    print("last digit's end -> ext 0:", int(np.median(e[:, 48] - e[:, 47])), " ext 0 end -> ext 1:", int(np.median(e[:, 52] - e[:, 51])))
tot = a[:, 60] - a[:, 0]
print("item start -> digit 0:", int(np.median(a[:, 1] - a[:, 0])), "last digit's end -> item end:", int(np.median(a[:, 60] - a[:, 47 if "--dma" in sys.argv else 48])))
print("wave lifetime median", int(np.median(tot)), "p10", int(np.percentile(tot, 10)), "p90", int(np.percentile(tot, 90)))
