"""Readers for the committed golden fixtures (tests/golden/*.npz, *.txt)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_g1():
    z = np.load(os.path.join(GOLDEN, "g1_matrix.npz"))
    cases = []
    for line in z["index"]:
        cid, si, M, K, N, dt, kind, aq, ub, iscale, qkv = str(line).split("|")
        key = f"c{cid}"
        c = dict(id=int(cid), M=int(M), K=int(K), N=int(N), dt=dt, kind=kind, act_quant=aq,
                 use_bias=bool(int(ub)), input_scale=float(iscale), qkv_size=[int(v) for v in qkv.split(",")],
                 x=z[key + "_x"], out=z[key + "_out"], xq=z[key + "_xq"], acc=z[key + "_acc"],
                 wq=z[f"wq_{si}_{'qkv' if kind == 'qkv' else 'plain'}"], W=z[f"W{si}"],
                 bias=z[f"b{si}"] if int(ub) else None)
        if kind == "qkv":
            c["qkv_scales"] = z[key + "_qkv_scales"]
        else:
            c["dequant_scale"] = float(z[key + "_dequant_scale"])
        if key + "_quant_scale" in z.files:
            c["quant_scale"] = float(z[key + "_quant_scale"])
        cases.append(c)
    return cases


def load_g2():
    z = np.load(os.path.join(GOLDEN, "g2_edges.npz"))
    cases = []
    for line in z["index"]:
        name, dt, kind, aq, ub, iscale = str(line).split("|")
        c = dict(id=name, dt=dt, kind=kind, act_quant=aq, use_bias=bool(int(ub)), input_scale=float(iscale),
                 x=z[name + "_x"], out=z[name + "_out"], xq=z[name + "_xq"], acc=z[name + "_acc"],
                 wq=z[name + "_wq"], dequant_scale=float(z[name + "_dequant_scale"]),
                 bias=z["b"] if int(ub) else None)
        if name + "_quant_scale" in z.files:
            c["quant_scale"] = float(z[name + "_quant_scale"])
        cases.append(c)
    return cases


def load_g2_bigk():
    z = np.load(os.path.join(GOLDEN, "g2_bigk.npz"))
    return {k: z[k] for k in z.files}


def load_g3():
    z = np.load(os.path.join(GOLDEN, "g3_from_float.npz"))
    return z, [str(s).split("|") for s in z["index"]]


def load_g4():
    cases = []
    with open(os.path.join(GOLDEN, "g4_config_hashes.txt")) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            (name, kind, M, K, N, dt, aq, ub, stream, ds, qs, h_xq, h_acc, h_out, samples) = line.rstrip("\n").split("|")
            smp = []
            for s in samples.split(";"):
                i, a, o = s.split(":")
                smp.append((int(i), int(a), float(o)))
            cases.append(dict(id=name, kind=kind, M=int(M), K=int(K), N=int(N), dt=dt, act_quant=aq,
                              use_bias=bool(int(ub)), stream=int(stream), dequant_scale=float(ds),
                              quant_scale=float(qs), sha_xq=h_xq, sha_acc=h_acc, sha_out=h_out, samples=smp))
    return cases


def g4_inputs(c):
    """Regenerate a G4 case's inputs from the build-owned RNG (same bytes anywhere)."""
    import detrng
    from oracle import w8a8 as O
    ci = c["stream"]
    wq = detrng.int8_uniform(41, ci, (c["N"], c["K"]))
    bias = (detrng.normal(42, ci, (c["N"],)) * np.float32(0.5)).astype(np.float32) if c["use_bias"] else None
    x = O.round_to(detrng.act_like(43, ci, (c["M"], c["K"]), scale=40.0), c["dt"])
    return x, wq, bias


def load_g6():
    z = np.load(os.path.join(GOLDEN, "g6_smooth.npz"))
    cases = []
    for line in z["index"]:
        ci, kind, mt, alpha, dt, nfc = str(line).split("|")
        k = f"c{ci}"
        cases.append(dict(id=int(ci), kind=kind, model_type=mt, alpha=float(alpha), dt=dt, ln_w=z[k + "_ln_w"], ln_w_out=z[k + "_ln_w_out"],
                          ln_b=z[k + "_ln_b"] if kind == "layernorm" else None, ln_b_out=z[k + "_ln_b_out"] if kind == "layernorm" else None,
                          act=z[k + "_act"], fcs=[z[k + f"_fc{j}"] for j in range(int(nfc))], fcs_out=[z[k + f"_fc{j}_out"] for j in range(int(nfc))]))
    return cases


def load_g7():
    z = np.load(os.path.join(GOLDEN, "g7_block.npz"))
    H, heads, inter, B, S = [int(v) for v in z["dims"]]
    cases = []
    for line in z["index"]:
        ci, a, b, c, d = str(line).split("|")
        cases.append(dict(id=int(ci), qc={"qkv": a, "out": b, "fc1": c, "fc2": d}, y=z[f"c{ci}_y"]))
    return dict(H=H, heads=heads, inter=inter, B=B, S=S, eps=float(z["eps"]), x=z["x"], y_float=z["y_float"], scales=z["scales"],
                W={k[2:]: z[k] for k in z.files if k.startswith("W_")}, cases=cases)


def load_g8():
    """N1 fixtures (tests/golden/make_golden_n1.py): LayerNormQ, dq_add_layernorm_q_py, scale-folded RMSNorm + round."""
    z = np.load(os.path.join(GOLDEN, "g8_n1.npz"))
    cases = []
    for line in z["index"]:
        kind, ci, dt, H, M, eps = str(line).split("|")
        key = f"{kind}_{ci}"
        c = dict(kind=kind, id=key, dt=dt, H=int(H), M=int(M), eps=float(eps))
        for f in z.files:
            if f.startswith(key + "_"):
                c[f[len(key) + 1:]] = z[f]
        cases.append(c)
    return cases
