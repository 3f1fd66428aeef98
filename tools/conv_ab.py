#!/usr/bin/env python3
"""A/B micro-benchmark of the fused conv kernel across library builds (guide rule: compare variants on one
device, random data). Build variants in the container first:
    python tools/conv_ab.py --build base: stamps:CDDPM_STAMPS accsilu:CDDPM_ACCURATE_SILU
then on the GPU box:
    python tools/conv_ab.py --run base stamps accsilu
Each variant runs in its own process (CDDPM_LIB selects the .so); shapes are the UNet's layers at B=64."""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "conditioned-diffusion-models-uad_amd"
CSRC = os.path.join(ROOT, PKG, "csrc")

# name, C0, C1, Cout, k, H, W, coef, silu, up, res_mode, skipC
SHAPES = [
    ("conv1 128>128 @128", 128, 0, 128, 3, 128, 128, 1, 1, 0, 0, 0),
    ("conv2 128>128 @128 +res", 128, 0, 128, 3, 128, 128, 1, 1, 0, 1, 0),
    ("plain 128>128 @128 nocoef", 128, 0, 128, 3, 128, 128, 0, 0, 0, 0, 0),
    ("coef only 128>128 @128", 128, 0, 128, 3, 128, 128, 1, 0, 0, 0, 0),
    ("silu only 128>128 @128", 128, 0, 128, 3, 128, 128, 0, 1, 0, 0, 0),
    ("up conv1 256>256 @128", 256, 0, 256, 3, 128, 128, 1, 1, 1, 0, 0),
    ("cat conv1 512>256 @64", 256, 256, 256, 3, 64, 64, 1, 1, 0, 0, 0),
    ("conv2 256>256 @64 +skip512", 256, 0, 256, 3, 64, 64, 1, 1, 0, 0, 512),
    ("conv2 256>256 @64 noskip", 256, 0, 256, 3, 64, 64, 1, 1, 0, 0, 0),
    ("conv2 128>128 @128 +skip384", 128, 0, 128, 3, 128, 128, 1, 1, 0, 0, 384),
    ("conv 256>256 @32", 256, 0, 256, 3, 32, 32, 1, 1, 0, 0, 0),
    ("cat conv1 384>128 @128", 256, 128, 128, 3, 128, 128, 1, 1, 0, 0, 0),
]
PHASES = ["prologue", "patch stage", "weight stage", "mfma compute", "chunk fold", "epilogue"]
FINE = ["chunk: wait vmcnt", "chunk: barrier A", "patch transform+store", "chunk: barrier B", "stage end: wait vmcnt", "stage end: barrier",
        "issue + MFMA + fold", "prologue + epilogue"]
EPI = ["prologue", "main loop", "residual requests", "barrier", "transpose 0", "stores+stats 0", "transpose 1", "stores+stats 1"]
PHASES8 = ["prologue", "chunk barrier", "patch store", "weight store", "stage barrier", "compute", "fold", "epilogue"]


def child(B, iters):
    lib = importlib.import_module(PKG + "._lib").load_library()
    eng = importlib.import_module(PKG + ".engine")
    e = eng.CddpmEngine(timesteps=10, max_batch=1, max_h=32, max_w=32)
    out = []
    for (name, C0, C1, Cout, k, H, W, coef, silu, up, res, sk) in SHAPES:
        ms = C.c_double()
        st = (C.c_uint64 * 64)()
        rc = lib.cddpm_op_conv_bench(e._h, C0, C1, Cout, k, B, H, W, coef, silu, up, res, sk, iters, C.byref(ms), st)
        assert rc == 0, lib.cddpm_last_error(e._h)
        flops = 2.0 * B * H * W * Cout * ((C0 + C1) * k * k + sk)
        row = {"shape": name, "ms": ms.value, "tflops": flops / ms.value / 1e9}
        tot = [sum(st[w * 8 + i] for w in range(4)) for i in range(6)]
        if st[41]:
            row["clock_GHz"] = round(st[40] / st[41] * 0.1, 3)
        if os.environ.get("CDDPM_CONV_WS") == "1" and sum(st[:4]):
            row["ws_cycles"] = {"matrix work": int(st[0]), "matrix barrier": int(st[1]), "staging work": int(st[2]), "staging barrier": int(st[3])}
        if sum(st[48:56]) and sum(st[:8]):
            tt = [st[i] for i in range(8)]
            names = ("c:mfma k0", "c:frag reads k1", "c:patch entry", "c:mfma k1", "s:fold", "s:weight write", "s:requests", "s:frag prefetch")
            row["pp_fine"] = {k: round(v / max(1, sum(tt)), 3) for k, v in zip(names, tt)}
        if sum(st[48:56]):
            for g in (0, 1):
                tt = [st[48 + 4 * g + i] for i in range(4)]
                row["pp_group%d" % g] = {k: round(v / max(1, sum(tt)), 3) for k, v in zip(("compute", "staging", "wait after compute", "wait after staging"), tt)}
        if os.environ.get("AB_EPI") and sum(st[:32]):
            tt = [sum(st[w * 8 + i] for w in range(4)) for i in range(8)]
            row["fine"] = {p: round(t / max(1, sum(tt)), 4) for p, t in zip(EPI, tt)}
        elif os.environ.get("AB_FINE") and sum(st[:32]):
            tt = [sum(st[w * 8 + i] for w in range(4)) for i in range(8)]
            row["fine"] = {p: round(t / max(1, sum(tt)), 4) for p, t in zip(FINE, tt)}
        elif os.environ.get("AB_PHASES8") and sum(st[:16]):
            for wv in (0, 1):
                tt = [st[wv * 8 + i] for i in range(8)]
                row["wave%d" % (wv * 4)] = {p: round(t / max(1, sum(tt)), 3) for p, t in zip(PHASES8, tt)}
        elif sum(tot):
            row["phase_share"] = {p: round(t / sum(tot), 4) for p, t in zip(PHASES, tot)}
        out.append(row)
    print(json.dumps(out))


def main():
    if sys.argv[1] == "--build":
        b = importlib.import_module(PKG + ".build")
        for spec in sys.argv[2:]:
            tag, _, rest = spec.partition(":")
            defs, _, src = rest.partition("@")          # tag:DEF1,DEF2@path/to/alternative_conv_mfma.hip
            print(b.build_lib(force=True, defines=[d for d in defs.split(",") if d], tag=tag, conv_src=src))
    elif sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    elif sys.argv[1] == "--run":
        B = int(os.environ.get("AB_BATCH", "64"))
        rounds = int(os.environ.get("AB_ROUNDS", "2"))
        res = {}
        for r in range(rounds):          # interleaved rounds
            for tag in sys.argv[2:]:
                lib, _, opt = tag.partition("+")          # "cur+ws": library cur with CDDPM_CONV_WS=1
                env = dict(os.environ, CDDPM_LIB=os.path.join(CSRC, "libcddpm_hip.so" if lib == "prod" else f"libcddpm_hip_{lib}.so"))
                if opt in ("nb1", "nb2"):          # 256-cout workgroups (conv_x6.hip, NB = 2) off / on wherever the kernel can
                    env["CDDPM_NB2"] = "0" if opt == "nb1" else "force"
                if opt == "ws":
                    env["CDDPM_CONV_WS"] = "1"
                if opt in ("f32", "x6", "h3"):
                    env["CDDPM_CONV"] = opt
                if opt == "pp":
                    env["CDDPM_CONV_PP"] = "1"
                if opt in ("m16", "m32"):
                    env["CDDPM_M16"] = "1" if opt == "m16" else "0"
                if opt == "r4":
                    env["CDDPM_ROWS"] = "4"
                if opt == "zero":
                    env["CDDPM_BENCH_ZERO"] = "1"
                if opt in ("w4", "w8"):
                    env["CDDPM_CONV_WAVES"] = opt[1]
                o = subprocess.run([sys.executable, __file__, "--child", str(B), "5"], env=env, capture_output=True, text=True)
                if o.returncode != 0:
                    print(tag, "FAILED", o.stderr[-2000:])
                    continue
                res.setdefault(tag, []).append(json.loads(o.stdout.strip().splitlines()[-1]))
        for i, shp in enumerate(SHAPES):
            print(shp[0])
            for tag, runs in res.items():
                ms = [r[i]["ms"] for r in runs]
                tf = [r[i]["tflops"] for r in runs]
                line = f"   {tag:10s} ms min {min(ms):8.3f} med {sorted(ms)[len(ms) // 2]:8.3f}  TF max {max(tf):6.1f}"
                if "phase_share" in runs[0][i]:
                    line += "  " + json.dumps(runs[0][i]["phase_share"])
                if "clock_GHz" in runs[0][i]:
                    line += f"  clock {runs[0][i]['clock_GHz']} GHz"
                for wk in ("fine", "wave0", "wave4", "pp_group0", "pp_group1", "pp_fine"):
                    if wk in runs[0][i]:
                        line += "\n        " + wk + " " + json.dumps(runs[0][i][wk])
                if "ws_cycles" in runs[0][i]:
                    line += "  " + json.dumps(runs[0][i]["ws_cycles"])
                print(line)


if __name__ == "__main__":
    main()
