"""Parameter / buffer inventory of MDViT and BASE and a build-owned deterministic
generator for test weights.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The names and shapes restate what the reference's constructors register
(/root/reference/Models/Transformer/mdvit.py:484-645, base.py:350-455,
Models/Decoders.py:174-214,289-313, Models/Transformer/mpvit.py:81-124,229-318);
SURVEY.md Appendix D holds the dumped table this was checked against.
Only *unique* tensors are listed (the stage-level name for the shared cpe/crpe).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

EMBED_DIMS = (64, 128, 320, 512)
MLP_RATIOS = (8, 8, 4, 4)
NUM_HEADS = (8, 8, 8, 8)
NUM_LAYERS = (2, 2, 2, 2)
CRPE_WINDOW = ((3, 2), (5, 3), (7, 3))  # (window, head split)  mdvit.py:423
AUX_HIDDEN = 512                         # MLPDecoderFM(embed_dims, 1, 512)  mdvit.py:595-599
NUM_DOMAINS = 4


def _bn(spec, prefix, c):
    spec[prefix + ".weight"] = ("bn_w", (c,))
    spec[prefix + ".bias"] = ("bn_b", (c,))
    spec[prefix + ".running_mean"] = ("bn_rm", (c,))
    spec[prefix + ".running_var"] = ("bn_rv", (c,))
    spec[prefix + ".num_batches_tracked"] = ("bn_nbt", ())


def _stage(spec, prefix, C, r, heads, layers, sup, num_domains):
    Ch = C // heads
    spec[f"{prefix}.cpe.proj.weight"] = ("dwconv", (C, 1, 3, 3))
    spec[f"{prefix}.cpe.proj.bias"] = ("bias", (C,))
    for w, (win, split) in enumerate(CRPE_WINDOW):
        spec[f"{prefix}.crpe.conv_list.{w}.weight"] = ("dwconv", (split * Ch, 1, win, win))
        spec[f"{prefix}.crpe.conv_list.{w}.bias"] = ("bias", (split * Ch,))
    for i in range(layers):
        b = f"{prefix}.mhca_blks.{i}"
        spec[f"{b}.norm1.weight"] = ("ln_w", (C,))
        spec[f"{b}.norm1.bias"] = ("ln_b", (C,))
        spec[f"{b}.factoratt_crpe.qkv.weight"] = ("linear", (3 * C, C))
        spec[f"{b}.factoratt_crpe.qkv.bias"] = ("bias", (3 * C,))
        spec[f"{b}.factoratt_crpe.proj.weight"] = ("linear", (C, C))
        spec[f"{b}.factoratt_crpe.proj.bias"] = ("bias", (C,))
        if sup:
            hid = max(C // 2, 4)
            spec[f"{b}.factoratt_crpe.domain_layer.0.weight"] = ("da", (hid, num_domains))
            spec[f"{b}.factoratt_crpe.domain_layer.0.bias"] = ("bias", (hid,))
            spec[f"{b}.factoratt_crpe.domain_layer.2.weight"] = ("da", (C, hid))
            spec[f"{b}.factoratt_crpe.domain_layer.2.bias"] = ("bias", (C,))
        spec[f"{b}.norm2.weight"] = ("ln_w", (C,))
        spec[f"{b}.norm2.bias"] = ("ln_b", (C,))
        spec[f"{b}.mlp.fc1.weight"] = ("linear", (r * C, C))
        spec[f"{b}.mlp.fc1.bias"] = ("bias", (r * C,))
        spec[f"{b}.mlp.fc2.weight"] = ("linear", (C, r * C))
        spec[f"{b}.mlp.fc2.bias"] = ("bias", (C,))


def param_spec(model: str = "MDViT", adapt_method="Sup", num_domains: int = NUM_DOMAINS,
               embed_dims=EMBED_DIMS, mlp_ratios=MLP_RATIOS, num_heads=NUM_HEADS,
               num_layers=NUM_LAYERS, in_chans: int = 3, decoder_name: str = "MLPFM") -> "OrderedDict[str, tuple]":
    """name -> (kind, shape) for every unique parameter and buffer.  decoder_name: 'MLPFM' (peer heads that also take the
    main decoder's last feature, Decoders.py:289-339), 'MLP' (Decoders.py:239-286: the four encoder features only) or
    'Transformer' (mdvit.py:614-642: per domain a full transformer decoder without Domain Adapter, `debranchs.{d}.{0..4}`),
    'DeepLabV3' (Decoders.py:218-236 + Utils/_deeplab.py:115-166: ASPP heads on the last encoder feature)."""
    assert decoder_name in ("MLPFM", "MLP", "Transformer", "DeepLabV3")
    if model == "MDViT_DSN":
        return _dsn_spec(param_spec("MDViT", adapt_method, num_domains, embed_dims, mlp_ratios, num_heads, num_layers, in_chans, decoder_name), num_domains)
    if model == "BASE_DSN":        # base.py:515-700: BASE with the same per-domain norm banks
        return _dsn_spec(param_spec("BASE", adapt_method, num_domains, embed_dims, mlp_ratios, num_heads, num_layers, in_chans), num_domains)
    assert model in ("MDViT", "BASE")
    sup = adapt_method == "Sup"
    E = tuple(embed_dims)
    spec: "OrderedDict[str, tuple]" = OrderedDict()
    spec["stem.0.conv.weight"] = ("conv", (E[0] // 2, in_chans, 3, 3))
    _bn(spec, "stem.0.bn", E[0] // 2)
    spec["stem.1.conv.weight"] = ("conv", (E[0], E[0] // 2, 3, 3))
    _bn(spec, "stem.1.bn", E[0])
    for s in range(4):
        cin = E[0] if s == 0 else E[s - 1]
        p = f"patch_embed_stages.{s}.patch_conv"
        spec[f"{p}.dwconv.weight"] = ("dwconv", (cin, 1, 3, 3))
        spec[f"{p}.pwconv.weight"] = ("conv", (E[s], cin, 1, 1))
        _bn(spec, f"{p}.bn", E[s])
    for s in range(4):
        _stage(spec, f"mhsa_stages.{s}", E[s], mlp_ratios[s], num_heads[s], num_layers[s], sup, num_domains)
    spec["bridge.0.weight"] = ("conv", (E[3], E[3], 3, 3))
    spec["bridge.0.bias"] = ("bias", (E[3],))
    _bn(spec, "bridge.1", E[3])
    spec["bridge.3.weight"] = ("conv", (2 * E[3], E[3], 3, 3))
    spec["bridge.3.bias"] = ("bias", (2 * E[3],))
    _bn(spec, "bridge.4", 2 * E[3])
    dec_io = ((2 * E[3], E[3]), (E[3], E[2]), (E[2], E[1]), (E[1], E[0]))
    for j, (cin, cout) in enumerate(dec_io, start=1):
        spec[f"decoder{j}.conv_before.weight"] = ("conv", (cout, cin, 1, 1))
        spec[f"decoder{j}.conv_before.bias"] = ("bias", (cout,))
        spec[f"decoder{j}.conv_after.dwconv.weight"] = ("dwconv", (cout, 2, 3, 3))
        spec[f"decoder{j}.conv_after.pwconv.weight"] = ("conv", (cout, cout, 1, 1))
        _bn(spec, f"decoder{j}.conv_after.bn", cout)
        s = 4 - j
        _stage(spec, f"decoder{j}.mhsa_block", E[s], mlp_ratios[s], num_heads[s], num_layers[s], sup, num_domains)
    spec["finalconv.0.weight"] = ("conv", (1, E[0], 1, 1))
    spec["finalconv.0.bias"] = ("bias", (1,))
    if model == "MDViT" and decoder_name == "Transformer":
        for d in range(num_domains):
            for j, (cin, cout) in enumerate(dec_io):
                pre = f"debranchs.{d}.{j}"
                spec[f"{pre}.conv_before.weight"] = ("conv", (cout, cin, 1, 1))
                spec[f"{pre}.conv_before.bias"] = ("bias", (cout,))
                spec[f"{pre}.conv_after.dwconv.weight"] = ("dwconv", (cout, 2, 3, 3))
                spec[f"{pre}.conv_after.pwconv.weight"] = ("conv", (cout, cout, 1, 1))
                _bn(spec, f"{pre}.conv_after.bn", cout)
                s = 3 - j
                _stage(spec, f"{pre}.mhsa_block", E[s], mlp_ratios[s], num_heads[s], num_layers[s], False, num_domains)
            spec[f"debranchs.{d}.4.0.weight"] = ("conv", (1, E[0], 1, 1))
            spec[f"debranchs.{d}.4.0.bias"] = ("bias", (1,))
    elif model == "MDViT" and decoder_name == "DeepLabV3":
        for d in range(1, 5):
            a = f"debranch{d}.classifier.0"
            spec[f"{a}.convs.0.0.weight"] = ("conv", (256, E[3], 1, 1))
            _bn(spec, f"{a}.convs.0.1", 256)
            for i in (1, 2, 3):
                spec[f"{a}.convs.{i}.0.weight"] = ("conv", (256, E[3], 3, 3))
                _bn(spec, f"{a}.convs.{i}.1", 256)
            spec[f"{a}.convs.4.1.weight"] = ("conv", (256, E[3], 1, 1))
            _bn(spec, f"{a}.convs.4.2", 256)
            spec[f"{a}.project.0.weight"] = ("conv", (256, 5 * 256, 1, 1))
            _bn(spec, f"{a}.project.1", 256)
            spec[f"debranch{d}.classifier.1.weight"] = ("conv", (256, 256, 3, 3))
            _bn(spec, f"debranch{d}.classifier.2", 256)
            spec[f"debranch{d}.classifier.4.weight"] = ("conv", (1, 256, 1, 1))
            spec[f"debranch{d}.classifier.4.bias"] = ("bias", (1,))
    elif model == "MDViT":
        for d in range(1, 5):
            for q in range(1, 5):
                spec[f"debranch{d}.linear{q}.weight"] = ("conv", (AUX_HIDDEN, E[q - 1], 1, 1))
                spec[f"debranch{d}.linear{q}.bias"] = ("bias", (AUX_HIDDEN,))
            spec[f"debranch{d}.linear_fuse.0.weight"] = ("conv", (AUX_HIDDEN, 4 * AUX_HIDDEN + (64 if decoder_name == "MLPFM" else 0), 1, 1))
            spec[f"debranch{d}.linear_fuse.0.bias"] = ("bias", (AUX_HIDDEN,))
            _bn(spec, f"debranch{d}.linear_fuse.1", AUX_HIDDEN)
            spec[f"debranch{d}.linear_out.weight"] = ("conv", (1, AUX_HIDDEN, 1, 1))
            spec[f"debranch{d}.linear_out.bias"] = ("bias", (1,))
    return spec


# ---------------------------------------------------------------------------------------------
# MDViT_DSN (mdvit.py:735-960): the MDViT graph with every trunk BatchNorm / LayerNorm replaced by a ModuleList of
# num_domains norms indexed by int(d) (Conv2d_BN_M :23-70, DWConv2d_BN_M :127-179, SerialBlock_adapt_M :364-412,
# bridge_norms{1,2} :815-820, Decoders.DWConv2d_BN_M :66-118).  Its parameter names follow from MDViT's by a renaming.
# ---------------------------------------------------------------------------------------------
_DSN_RULES = (
    (r"^stem\.0\.conv\.", "stem_1.conv.", False), (r"^stem\.0\.bn\.", "stem_1.bns.{d}.", True),
    (r"^stem\.1\.conv\.", "stem_2.conv.", False), (r"^stem\.1\.bn\.", "stem_2.bns.{d}.", True),
    (r"\.patch_conv\.bn\.", ".patch_conv.bns.{d}.", True),
    (r"\.norm1\.", ".norm1s.{d}.", True), (r"\.norm2\.", ".norm2s.{d}.", True),
    (r"^bridge\.0\.", "bridge_conv1.", False), (r"^bridge\.1\.", "bridge_norms1.{d}.", True),
    (r"^bridge\.3\.", "bridge_conv2.", False), (r"^bridge\.4\.", "bridge_norms2.{d}.", True),
    (r"\.conv_after\.bn\.", ".conv_after.bns.{d}.", True),
)


def dsn_name(canonical: str, d: int):
    """MDViT parameter name -> (MDViT_DSN name for domain d, is it domain specific?)"""
    import re as _re
    for pat, rep, per_domain in _DSN_RULES:
        if _re.search(pat, canonical):
            return _re.sub(pat, rep.format(d=d), canonical), per_domain
    return canonical, False


def _dsn_spec(mdvit_spec, num_domains):
    spec = OrderedDict()
    for name, ks in mdvit_spec.items():
        _, per_domain = dsn_name(name, 0)
        for d in range(num_domains if per_domain else 1):
            spec[dsn_name(name, d)[0]] = ks
    return spec


BUFFER_KINDS = ("bn_rm", "bn_rv", "bn_nbt")


def is_buffer(kind: str) -> bool:
    return kind in BUFFER_KINDS


# ---------------------------------------------------------------------------------------------
# deterministic generator (splitmix64 counter hash -> uniform), independent of any library RNG
# ---------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform_pm1(seed: int, stream: int, n: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream)], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + base)
    return (bits >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def _stream_id(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def make_params(seed: int = 0, **spec_kwargs) -> "OrderedDict[str, np.ndarray]":
    """Deterministic O(1)-activation test weights (NOT the reference's init scheme).

    Scales are chosen so activations stay O(1) through the network (SURVEY.md 7.3:
    reference-style random init saturates BCE / explodes in eval mode)."""
    spec = param_spec(**spec_kwargs)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (kind, shape) in spec.items():
        n = int(np.prod(shape)) if len(shape) else 1
        u = uniform_pm1(seed, _stream_id(name), n)
        if kind in ("conv", "dwconv"):
            fan_in = int(np.prod(shape[1:]))
            v = u * math.sqrt(3.0 / fan_in)              # unit-gain uniform
        elif kind == "linear":
            v = u * math.sqrt(3.0 / shape[1])
        elif kind == "da":
            v = u * (1.5 if shape[1] <= 8 else 3.0 / math.sqrt(shape[1]))
        elif kind == "bias":
            v = u * 0.1
        elif kind in ("ln_w", "bn_w"):
            v = 1.0 + 0.5 * u
        elif kind in ("ln_b", "bn_b"):
            v = 0.1 * u
        elif kind == "bn_rm":
            v = 0.1 * u
        elif kind == "bn_rv":
            v = 1.0 + 0.5 * u
        elif kind == "bn_nbt":
            out[name] = np.zeros((), dtype=np.int64)
            continue
        else:  # pragma: no cover
            raise KeyError(kind)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def alias_map(model: str = "MDViT", num_layers=NUM_LAYERS, decoder_name: str = "MLPFM", num_domains: int = NUM_DOMAINS) -> dict:
    """alias state_dict key -> unique key (shared cpe/crpe registered under every block;
    mdvit.py:426-435)."""
    amap = {}
    stages = [(f"mhsa_stages.{s}", s) for s in range(4)] + [(f"decoder{j}.mhsa_block", 4 - j) for j in range(1, 5)]
    if decoder_name == "Transformer":
        stages += [(f"debranchs.{d}.{j}.mhsa_block", 3 - j) for d in range(num_domains) for j in range(4)]
    for st, s in stages:
        for i in range(num_layers[s]):
            for t in ("weight", "bias"):
                amap[f"{st}.mhca_blks.{i}.cpe.proj.{t}"] = f"{st}.cpe.proj.{t}"
                for w in range(3):
                    amap[f"{st}.mhca_blks.{i}.factoratt_crpe.crpe.conv_list.{w}.{t}"] = f"{st}.crpe.conv_list.{w}.{t}"
    return amap
