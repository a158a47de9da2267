"""CPU-only checks: the C-ABI library loads and exports every declared symbol (no compute calls),
the module surface mirrors the reference's, the product never routes through the oracle, the
data-parallel gradient reducer is correct over gloo with world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from mdvit_amd import _lib
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mdvit_hip.h but not exported"
    assert lib.mdvit_version() == 1
    assert lib.mdvit_factoratt_ws_bytes(2, 256, 64, 8) > 0
    assert lib.mdvit_factoratt_ws_bytes(2, 256, 65, 8) == 0


def test_bad_arguments_return_error_codes_without_gpu():
    """argument validation happens before any HIP call, so it can be exercised on a CPU-only box"""
    import ctypes as C
    from mdvit_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.M, d.N, d.K = 0, 4, 4
    assert lib.mdvit_gemm_f32(C.byref(d), None) == 1          # MDVIT_E_SHAPE
    assert b"gemm" in lib.mdvit_last_error()
    assert lib.mdvit_stemconv_fwd(None, None, None, 1, 8, 8, 4, 32, None) == 1
    assert b"in_chans" in lib.mdvit_last_error()


def test_launch_sampler_is_off_and_empty_without_launches():
    """mdvit_gemm_sampler / _read (round 6, bench.py's roofline): arming and reading without a launch touches no GPU state -- an empty table ends at index 0 -- and a
    rejected GEMM call (bad shape) is not counted"""
    import ctypes as C
    from mdvit_amd import _lib
    lib = _lib.load()
    assert lib.mdvit_gemm_sampler(None, 1) == 0
    d = _lib.GemmDesc()
    d.M, d.N, d.K = 0, 4, 4
    assert lib.mdvit_gemm_f32(C.byref(d), None) == 1
    nm, seen, timed, ms = C.create_string_buffer(160), C.c_int64(), C.c_int64(), C.c_double()
    assert lib.mdvit_gemm_sampler_read(0, nm, 160, C.byref(seen), C.byref(timed), C.byref(ms)) != 0
    assert lib.mdvit_gemm_sampler(b"gemm_tn_kernel<128, 128, 2, true, false, false, false>", 4) == 0
    assert lib.mdvit_gemm_sampler(None, 0) == 0          # off
    assert lib.mdvit_gemm_sampler_read(0, nm, 160, C.byref(seen), C.byref(timed), C.byref(ms)) != 0


def test_bf16_stored_operands_are_validated_before_any_launch():
    """MdvitGemmDesc.a_bf16 / b_bf16 (the mixed mode's saved hidden activations): a layout of the weight-gradient (TN, precision 1) kernel only -- any other
    product, both operands at once, or a leading dimension that breaks the 8-byte quads is refused with MDVIT_E_SHAPE before anything touches a GPU."""
    import ctypes as C
    from mdvit_amd import _lib
    lib = _lib.load()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)

    def desc(**kw):
        d = _lib.GemmDesc()
        d.A, d.B, d.C = p, p, p
        d.M, d.N, d.K = 8, 8, 8
        d.lda, d.ldb, d.ldc = 8, 8, 8
        d.trans_a, d.trans_b, d.precision = 1, 0, 1
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    d = desc(a_bf16=1, trans_a=0, trans_b=1)                       # NT: not the weight-gradient layout
    assert lib.mdvit_gemm_f32(C.byref(d), None) == 1 and b"bf16-stored" in lib.mdvit_last_error()
    d = desc(a_bf16=1, b_bf16=1)                                   # both operands
    assert lib.mdvit_gemm_f32(C.byref(d), None) == 1 and b"bf16-stored" in lib.mdvit_last_error()
    d = desc(b_bf16=1, ldb=10)                                     # quads would straddle (the general alignment check catches it first)
    assert lib.mdvit_gemm_f32(C.byref(d), None) == 3 and b"leading dimensions" in lib.mdvit_last_error()
    d = desc(a_bf16=1, precision=0)                                # fp32 mode: the general template
    assert lib.mdvit_gemm_f32(C.byref(d), None) == 1 and b"bf16-stored" in lib.mdvit_last_error()
    assert lib.mdvit_gemm_tn_grid_order(2) == 0 and lib.mdvit_gemm_tn_grid_order(-1) == 0


def test_weight_gradient_planner_without_gpu():
    """mdvit_gemm_plan on weight-gradient (TN, bf16x3) descriptors -- host-side logic only: the K-split leaves every workgroup at least 8 slabs of 32 tokens,
    the launch has ~1.5 workgroups per CU in total when K allows it, a bf16-stored operand does not change the plan, and a tiny K is not split at all."""
    import ctypes as C
    from mdvit_amd import _lib
    lib = _lib.load()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)

    def plan(M, N, K, **kw):
        d = _lib.GemmDesc()
        d.A, d.B, d.C = p, p, p
        d.M, d.N, d.K = M, N, K
        d.lda, d.ldb, d.ldc = M, N, N
        d.trans_a, d.trans_b, d.precision, d.allow_split = 1, 0, 1, 1
        for k, v in kw.items():
            setattr(d, k, v)
        tm, tn, sp = C.c_int32(), C.c_int32(), C.c_int32()
        assert lib.mdvit_gemm_plan(C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp)) == 0
        return tm.value, tn.value, sp.value
    for (M, N, K) in ((1024, 128, 65536), (64, 512, 262144), (1280, 320, 16384), (512, 2048, 4096), (192, 64, 2097152)):
        tm, tn, sp = plan(M, N, K)
        tiles = -(-M // tm) * -(-N // tn)
        assert sp >= 1 and K // sp >= 256, (M, N, K, tm, tn, sp)              # >= 8 slabs of 32 tokens per workgroup
        assert 256 <= tiles * sp <= 768, (M, N, K, tm, tn, sp)                # ~384 workgroups in total
        assert plan(M, N, K, b_bf16=1) == (tm, tn, sp) and plan(M, N, K, a_bf16=1) == (tm, tn, sp)
    assert plan(128, 128, 200)[2] == 1                                        # fewer than 8 slabs: no split
    assert plan(128, 128, 65536, allow_split=0)[2] == 1


def test_block_roofline_bounds_are_what_the_bench_line_divides_by():
    """tools/block_roofline.py (the denominator of `block_bs32` in the bench line): per encoder stage at bs=32 the operator-sum bound of one SerialBlock_adapt
    forward + backward (every operator's max(bytes / 8 TB/s, flops / MFMA roof), fp32 storage), the STRICT bound without any T x hidden traffic (fused MLP), and
    SURVEY 8(d)'s whole-block fused-bf16 bound.  Pinned: strict <= operator-sum, SURVEY's figure below both, and the stage-0 values DESIGN.md quotes."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("block_roofline", os.path.join(root, "tools", "block_roofline.py"))
    br = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(br)
    B = 32
    for C, r, N in ((64, 8, 128 * 128), (128, 8, 64 * 64), (320, 4, 32 * 32), (512, 4, 16 * 16)):
        T, Hd = B * N, C * r
        f, b = br.block_ops(T, C, Hd, 8)
        fs, bs = br.block_ops_strict(T, C, Hd, 8)
        bf, bb = br.bound_seconds(f, "bf16x3")[0], br.bound_seconds(b, "bf16x3")[0]
        sf, sb = br.bound_seconds(fs, "bf16x3")[0], br.bound_seconds(bs, "bf16x3")[0]
        vf, vb = br.survey_bound(B, N, C, r, 8)
        assert 0 < sf <= bf and 0 < sb <= bb, (C, sf, bf, sb, bb)
        assert vf < sf and vb < sb, (C, vf, sf, vb, sb)
        assert not any(name.startswith("fc") for name, *_ in fs + bs)
        if C == 64:
            assert abs(bf * 1e3 - 0.604) < 2e-3 and abs(bb * 1e3 - 1.225) < 2e-3, (bf, bb)          # the "bound" column of profiles/r03*_block_roofline_bs32.txt
            assert abs(vf * 1e6 - 67) < 1 and abs(vb * 1e6 - 201) < 1, (vf, vb)


def test_pmc_summary_applies_the_gfx950_corrections(tmp_path):
    """tools/pmc_summary.py on two synthetic rocprofv3 --pmc passes: counters are KiB, averaged per launch of a kernel name (template arguments kept,
    argument list and namespace dropped), HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- the gfx950 read-side correction of MI355X_MICROARCH.md;
    this file is what bench.py reads for `roofline.traffic`."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = "Kernel_Name,Counter_Name,Counter_Value\n"
    k = '"void (anonymous namespace)::gemm_f32_kernel<64, 64, 2, 2, false, true, 0, true>((anonymous namespace)::GemmArgs)"'
    for d, counter, vals in (("fetch", "FETCH_SIZE", (100.0, 300.0)), ("write", "WRITE_SIZE", (50.0, 70.0))):
        os.makedirs(tmp_path / d / "x")
        with open(tmp_path / d / "x" / "pmc_counter_collection.csv", "w") as f:
            f.write(hdr + "".join(f"{k},{counter},{v}\n" for v in vals) + f'"other_kernel(int)",{counter},8.0\n')
    out = tmp_path / "traffic.json"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(tmp_path / "fetch"), str(tmp_path / "write"), str(out)],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    ks = json.load(open(out))["kernels"]
    g = ks["gemm_f32_kernel<64, 64, 2, 2, false, true, 0, true>"]
    assert g["launches_sampled"] == 2 and g["fetch_kib_raw_per_launch"] == 200.0 and g["write_kib_per_launch"] == 60.0
    assert g["hbm_bytes_per_launch"] == (2 * 200 + 60) * 1024
    assert ks["other_kernel"]["hbm_bytes_per_launch"] == (2 * 8 + 8) * 1024


def test_side_stream_hold_bookkeeping_without_gpu(monkeypatch):
    """ops._trim_side_groups (the bound on what the weight-gradient stream keeps allocated), on stand-in events and streams: groups whose event has completed
    are dropped first, oldest first; past the bound the OWNING stream is made to wait for the oldest group's event before that group is dropped; the byte
    count follows; nothing is dropped while under the bound and incomplete."""
    from mdvit_amd import ops

    class Ev:
        def __init__(self, done):
            self.done = done

        def query(self):
            return self.done

    class St:
        def __init__(self):
            self.waited = []

        def wait_event(self, ev):
            self.waited.append(ev)

    main, branch = St(), St()
    e = [Ev(True), Ev(False), Ev(False), Ev(False)]
    groups = [(e[0], main, ["a"], 10), (e[1], main, ["b"], 20), (e[2], branch, ["c"], 30), (e[3], main, ["d"], 5)]
    monkeypatch.setattr(ops, "_side_groups", list(groups))
    monkeypatch.setattr(ops, "_side_held", [65])
    ops._trim_side_groups(100)                       # under the bound: only the completed head goes
    assert [g[2] for g in ops._side_groups] == [["b"], ["c"], ["d"]] and ops._side_held[0] == 55 and not main.waited and not branch.waited
    ops._trim_side_groups(40)                        # 55 > 40: the oldest group's owner waits for its event, 35 <= 40 stops it
    assert [g[2] for g in ops._side_groups] == [["c"], ["d"]] and ops._side_held[0] == 35 and main.waited == [e[1]] and not branch.waited
    ops._trim_side_groups(1)                         # the branch stream's group waits on ITS stream; the last group on main's
    assert ops._side_groups == [] and ops._side_held[0] == 0 and branch.waited == [e[2]] and main.waited == [e[1], e[3]]


def test_block_entry_routing_predicates_without_gpu():
    """ops.block_entry_ok / _lin_rc_ok / _mlp_rc16_ok (pure host logic): which SerialBlock_adapt configurations take the one-call C entry, and which of their
    layers the streaming kernels.  Parity (bf16x3) mode: every encoder width; the bf16 ("mixed") mode: the C <= 128 blocks keep the bf16x3 register-chained
    kernels (with bf16-stored h / du at C = 128), the MFMA-bound C >= 320 blocks go to the operator path and its single-plane GEMMs."""
    from mdvit_amd import ops
    prev = ops.gemm_precision()

    def params(C, hidden):
        t = lambda *sh: torch.zeros(*sh)
        return [t(C, 1, 3, 3), t(C), t(C), t(C), t(3 * C, C), t(3 * C), t(C // 4, 1, 3, 3), t(C // 4), t(3 * C // 8, 1, 5, 5), t(3 * C // 8), t(3 * C // 8, 1, 7, 7),
                t(3 * C // 8), None, None, None, None, t(C, C), t(C), t(C), t(C), t(hidden, C), t(hidden), t(C, hidden), t(C)]
    try:
        ops.set_gemm_precision("bf16x3")
        for C, r in ((64, 8), (128, 8), (320, 4), (512, 4)):
            assert ops.block_entry_ok(C, C * r, params(C, C * r)), C
        assert not ops.block_entry_ok(66, 528, params(64, 512))                       # C % 4
        assert ops._lin_rc_ok(65536, 192, 64) and ops._lin_rc_ok(4096, 128, 128)          # short-K Linear layers of the C = 64 / 128 blocks: the streaming kernel
        assert not ops._lin_rc_ok(4096, 960, 320) and not ops._lin_rc_ok(512, 192, 64)    # K = 320: the tiled GEMM; fewer than 1024 rows: not worth a launch shape of its own
        ops.set_gemm_precision("bf16")
        assert ops.block_entry_ok(64, 512, params(64, 512)) and ops.block_entry_ok(128, 1024, params(128, 1024))
        assert not ops.block_entry_ok(320, 1280, params(320, 1280)) and not ops.block_entry_ok(512, 2048, params(512, 2048))
        ops.set_gemm_precision("fp32")
        assert not ops._lin_rc_ok(65536, 192, 64)                                     # the fp32 mode has no plane kernels
    finally:
        ops.set_gemm_precision(prev)


def test_ctypes_mirrors_have_the_layout_of_the_public_header(tmp_path):
    """include/mdvit_hip.h is plain C: compile it with gcc and compare sizeof and every field offset of the five ABI structs with their ctypes mirrors in
    mdvit_amd/_lib.py -- a field added on one side only (this round: a_bf16 / b_bf16, store_bf16, ln_accumulate, the plane pointers) would otherwise
    shift everything behind it silently."""
    import ctypes as C, shutil, subprocess
    from mdvit_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = (("MdvitGemmDesc", _lib.GemmDesc), ("MdvitPlaneGemmDesc", _lib.PlaneGemmDesc), ("MdvitBlockDesc", _lib.BlockDesc),
             ("MdvitBlockGrads", _lib.BlockGrads), ("MdvitBlockStreams", _lib.BlockStreams), ("MdvitDaMany", _lib.DaMany), ("MdvitDaManyGrads", _lib.DaManyGrads))
    lines = ['#include <stdio.h>', '#include "mdvit_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'    printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _t in cls._fields_:
            lines.append(f'    printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['    return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]          # (a ctypes field the header does not have fails HERE)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, timeout=30).stdout.strip().splitlines())
    for cname, cls in pairs:
        assert int(out[cname]) == C.sizeof(cls), (cname, out[cname], C.sizeof(cls))
        for fname, _t in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname, out[f"{cname}.{fname}"], getattr(cls, fname).offset)


def test_the_package_imports_in_a_tree_without_the_library_and_its_use_fails_loudly(tmp_path):
    """a fresh checkout: `python -m mdvit_amd.build` and __graft_entry__.build() import the package BEFORE the library exists -- importing must not need the .so
    (process-wide configuration calls are queued, _lib.on_load), using it must raise (no CPU or PyTorch fallback), and the queued calls run at the first load"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import mdvit_amd\n"
        "from mdvit_amd import _lib, ops\n"
        "assert _lib._lib is None and len(_lib._on_load) >= 1, (_lib._lib, len(_lib._on_load))\n"
        "try:\n"
        "    _lib.load()\n"
        "except _lib.MdvitHipError as e:\n"
        "    assert 'is missing' in str(e) and 'no CPU or PyTorch fallback' in str(e)\n"
        "else:\n"
        "    raise SystemExit('load() of a missing library did not raise')\n"
        "try:\n"
        "    ops.call('mdvit_gemm_pm_config', 0)\n"
        "except _lib.MdvitHipError:\n"
        "    pass\n"
        "else:\n"
        "    raise SystemExit('a library call ran without the library')\n"
        "_lib.LIB_PATH = %r\n"
        "lib = _lib.load()\n"
        "assert len(_lib._on_load) == 0\n"
        "print('ok')\n") % (root, os.path.join(root, "mdvit_amd", "lib", "libmdvit_hip.so"))
    env = dict(os.environ, MDVIT_HIP_LIB=str(tmp_path / "not_built" / "libmdvit_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-3000:])


def test_a_stale_library_neither_breaks_the_import_nor_the_build_command_and_its_use_says_rebuild(tmp_path):
    """a git-ignored .so left over from an EARLIER tree lacks the newest entry points.  Importing the package (which `python -m mdvit_amd.build`,
    the command that would rebuild it, does first) must not dlopen it; load() must name the problem and the remedy"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "stale.c"
    src.write_text("const char* mdvit_last_error(void) { return \"\"; }\nint mdvit_version(void) { return 1; }\nint mdvit_gemm_f32(void* d, void* s) { return 0; }\n")
    so = tmp_path / "libmdvit_hip.so"
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True, timeout=60)
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import mdvit_amd\n"
        "import mdvit_amd.build\n"
        "from mdvit_amd import _lib\n"
        "assert _lib._lib is None, 'importing the package loaded the library'\n"
        "try:\n"
        "    _lib.load()\n"
        "except _lib.MdvitHipError as e:\n"
        "    assert 'stale' in str(e) and 'python -m mdvit_amd.build' in str(e), str(e)\n"
        "else:\n"
        "    raise SystemExit('load() of a stale library did not raise')\n"
        "assert _lib._lib is None and len(_lib._on_load) >= 1\n"
        "print('ok')\n") % (root,)
    env = dict(os.environ, MDVIT_HIP_LIB=str(so))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-3000:])


def test_ctypes_prototypes_have_the_arity_of_the_header_declarations():
    """every function mdvit_amd/_lib.py gives argtypes to: as many arguments as its declaration in include/mdvit_hip.h (a parameter added to one side only
    would pass garbage for everything behind it); and every declared function that the Python side calls has a prototype"""
    import re
    from mdvit_amd import _lib
    with open(_lib.HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\s*\*)\s+(mdvit_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    assert len(decl) >= 90, len(decl)
    bad = [(n, len(sig), decl.get(n)) for n, sig in _lib._SIGS.items() if decl.get(n) != len(sig)]
    assert not bad, f"argtypes vs header declarations (name, ctypes, header): {bad}"


def test_bench_bounds_the_host_run_ahead_by_batch(monkeypatch):
    """bench.py --max-inflight: two steps of run-ahead below batch 16, one from batch 16 up (every step of run-ahead keeps one more step's cross-stream
    tensors in the reserved pool; the host needs 22 ms for a 250 ms step there), an explicit value wins"""
    import importlib, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    for argv, want in ((["bench.py"], 2), (["bench.py", "--batch", "8"], 2), (["bench.py", "--batch", "16"], 1), (["bench.py", "--batch", "32"], 1),
                       (["bench.py", "--batch", "32", "--max-inflight", "3"], 3)):
        monkeypatch.setattr(sys, "argv", argv)
        assert bench.parse().max_inflight == want, argv


def test_ops_refuse_cpu_tensors():
    from mdvit_amd import ops, _lib
    with pytest.raises(_lib.MdvitHipError):
        ops.layer_norm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mdvit_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn
            assert "/root/reference" not in src, fn
    for fn in ("bench.py", "__graft_entry__.py"):
        p = os.path.join(ROOT, fn)
        if os.path.exists(p):
            assert "/root/reference" not in open(p).read().replace("os.path.isdir('/root/reference')", "")


def test_module_surface_matches_reference_inventory():
    import mdvit_amd
    from oracle.params import param_spec, alias_map
    m = mdvit_amd.MDViT(img_size=64, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                        num_domains=4, decoder_name="MLPFM")
    sd = m.state_dict()
    assert set(sd) == set(param_spec("MDViT", "Sup")) | set(alias_map())
    b = mdvit_amd.BASE(drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method=False)
    assert set(k for k in b.state_dict()) == set(param_spec("BASE", False)) | set(alias_map("BASE"))
    dsn = mdvit_amd.MDViT_DSN(img_size=64, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLPFM")
    assert set(dsn.state_dict()) == set(param_spec("MDViT_DSN", "Sup")) | set(alias_map())       # 852 unique + 128 aliases (mdvit.py:735-960)
    mlp = mdvit_amd.MDViT(img_size=64, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="MLP")
    spec_mlp = param_spec("MDViT", "Sup", decoder_name="MLP")
    assert set(mlp.state_dict()) == set(spec_mlp) | set(alias_map())
    assert tuple(mlp.debranch3.linear_fuse[0].weight.shape) == spec_mlp["debranch3.linear_fuse.0.weight"][1] == (512, 2048, 1, 1)
    assert mdvit_amd.MDViT_DSN(img_size=64, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", decoder_name="MLP").decoder_name == "MLP"
    tr = mdvit_amd.MDViT(img_size=64, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="Transformer")
    assert set(tr.state_dict()) == set(param_spec("MDViT", "Sup", decoder_name="Transformer")) | set(alias_map(decoder_name="Transformer"))
    assert not any("domain_layer" in k for k in tr.state_dict() if k.startswith("debranchs."))      # peers carry no adapter (mdvit.py:631)
    dl = mdvit_amd.MDViT(img_size=64, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4, decoder_name="DeepLabV3")
    assert set(dl.state_dict()) == set(param_spec("MDViT", "Sup", decoder_name="DeepLabV3")) | set(alias_map())
    bd = mdvit_amd.BASE_DSN(conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4)
    assert set(bd.state_dict()) == set(param_spec("BASE_DSN", "Sup")) | set(alias_map("BASE"))
    with pytest.raises(ValueError):
        mdvit_amd.MDViT(decoder_name="UperNet")
    with pytest.raises(NotImplementedError):
        mdvit_amd.MDViT_DSN(decoder_name="Transformer")
    # reference init scheme (mdvit.py:648-664)
    w = m.mhsa_stages[0].mhca_blks[0].mlp.fc1.weight
    assert abs(float(w.std()) - 0.02) < 0.003 and float(w.abs().max()) <= 2.0
    assert float(m.mhsa_stages[0].mhca_blks[0].mlp.fc1.bias.abs().max()) == 0
    dw = m.patch_embed_stages[1].patch_conv.dwconv.weight
    assert abs(float(dw.std()) - (2.0 / 9) ** 0.5) < 0.05


def test_transfuse_module_surface_matches_reference_inventory():
    """TransFuse_S_adapt built on the CPU (parameters only): the reference's 630 state_dict keys and shapes (TransFuse.py:182-226,
    torchvision's ResNet-34 names, DeiT's transformer.* names), 26.87 M parameters"""
    from mdvit_amd.transfuse import TransFuse_S_adapt
    from oracle.transfuse_ref import param_spec
    m = TransFuse_S_adapt(num_classes=1, drop_rate=0.2, pretrained=False, num_domains=4)
    sd, spec = m.state_dict(), param_spec()
    assert set(sd) == set(spec) and len(sd) == 630
    for k, (kind, shape) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in m.parameters()) == 26873877


def test_bench_gpus_n_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` typed as is (no torchrun around it): the parent starts the ranks through torch.distributed.run with a
    127.0.0.1 rendezvous and relays rank 0's JSON line; checked here without a GPU through the dry-run hook (gloo)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MDVIT_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       timeout=240, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"dryrun": True, "n_gpus": 2, "world": 2, "ranks_seen": 2, "steps": 3, "warmup": 1}


def test_synthetic_batches_follow_the_loader_contract():
    from mdvit_amd.synthetic import make_step_batches
    bs = make_step_batches(2, 64, rank=0)
    assert len(bs) == 4
    for d, (img, lab, sid) in enumerate(bs):
        assert img.shape == (2, 3, 64, 64) and img.dtype == torch.float32
        assert lab.shape == (2, 1, 64, 64) and set(lab.unique().tolist()) <= {0.0, 1.0}
        assert sid.tolist() == [d, d] and sid.dtype == torch.long
        assert -2.2 < float(img.min()) and float(img.max()) < 2.7
        assert 0.01 < float(lab.mean()) < 0.6
    again = make_step_batches(2, 64, rank=0)
    assert torch.equal(bs[1][0], again[1][0])
    other = make_step_batches(2, 64, rank=1)
    assert not torch.equal(bs[1][0], other[1][0])


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from mdvit_amd.parallel import GradBucketReducer, broadcast_parameters
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))
extra = torch.nn.Linear(5, 5)                 # parameters that never get a gradient (an unused aux head)
params = list(model.parameters()) + list(extra.parameters())
broadcast_parameters(model)
red = GradBucketReducer(params, bucket_bytes=4096)       # several buckets
assert len(red.buckets) > 2
g = torch.Generator().manual_seed(123)
X = torch.randn(8, 16, generator=g); Y = torch.randn(8, 3, generator=g)
xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
for step in range(2):
    red.zero_grad()
    # two backward sweeps accumulate; only the last one is armed (as in the two-sweep MDViT step)
    l1 = ((model(xs) - ys) ** 2).mean()
    l1.backward()
    l2 = (model(xs).abs()).mean()
    red.arm()
    l2.backward()
    red.finish()
    assert red.check_views()
ref = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))
ref.load_state_dict(model.state_dict())
tot = 0
for r in range(world):
    xr, yr = X[r * 4:(r + 1) * 4], Y[r * 4:(r + 1) * 4]
    tot = tot + (((ref(xr) - yr) ** 2).mean() + ref(xr).abs().mean()) / world
tot.backward()
for p, q in zip(model.parameters(), ref.parameters()):
    assert torch.allclose(p.grad, q.grad, atol=1e-6), (p.grad - q.grad).abs().max()
for p in extra.parameters():
    assert float(p.grad.abs().max()) == 0.0
# ---- fused accumulator: sweeps 1..n-1 folded by one multi-tensor add, the last sweep armed for all-reduce ----
from mdvit_amd.parallel import GradAccumulator
model2 = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))
model2.load_state_dict(ref.state_dict())
acc = GradAccumulator(list(model2.parameters()), bucket_bytes=4096)
for step in range(2):
    acc.zero()
    acc.begin_sweep(False); ((model2(xs) - ys) ** 2).mean().backward(); acc.end_sweep(False)
    acc.begin_sweep(True); model2(xs).abs().mean().backward(); acc.end_sweep(True)
for p, q in zip(model2.parameters(), ref.parameters()):
    assert torch.allclose(p.grad, q.grad, atol=1e-6), (p.grad - q.grad).abs().max()
# ---- the product path of the merged two-sweep step: gradient SINKS (weight gradients written straight into the buckets, never
# seen by autograd), `late` parameters (the domain adapters) in buckets of their own, every other bucket all-reduced right after
# the full sweep -- i.e. underneath the second (adapter-only) sweep -- and the late buckets after it
model3 = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))
model3.load_state_dict(ref.state_dict())
late = list(model3[2].parameters())                       # plays the domain adapters
acc3 = GradAccumulator(list(model3.parameters()), bucket_bytes=4096, late=late)
late_buckets = {{acc3.reducer._bucket_of[p] for p in late}}
assert all(acc3.reducer._bucket_of[p] not in late_buckets for p in model3.parameters() if all(p is not q for q in late))
sunk = [model3[0].weight, model3[4].weight]              # their "wgrad kernels" add into the sinks
for step in range(2):
    acc3.zero()
    l_aux = ((model3(xs) - ys) ** 2).mean(); l_uni = model3(xs).abs().mean()
    acc3.begin_sweep(False)
    (l_aux + l_uni).backward(retain_graph=True)           # the full sweep first ...
    for p in sunk:
        acc3.view_of(p).add_(p.grad); p.grad = None         # ... with these gradients delivered through the sinks
    acc3.end_sweep(False, remaining=late)
    assert acc3.overlapped_buckets == len(acc3.reducer.buckets) - len(late_buckets) > 0
    acc3.begin_sweep(True)                                # ... then the adapter-only sweep: minus their aux gradient
    for p, g_ in zip(late, torch.autograd.grad(l_aux, late)):
        p.grad = -g_
    acc3.end_sweep(True)
ref3 = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 3))
ref3.load_state_dict(ref.state_dict())
tot_aux = tot_uni = 0
for r in range(world):
    xr, yr = X[r * 4:(r + 1) * 4], Y[r * 4:(r + 1) * 4]
    tot_aux = tot_aux + ((ref3(xr) - yr) ** 2).mean() / world
    tot_uni = tot_uni + ref3(xr).abs().mean() / world
late3 = list(ref3[2].parameters())
for q in late3:
    q.requires_grad_(False)
tot_aux.backward(retain_graph=True)                       # the reference's order: aux with the adapters frozen, then uni into everything
for q in late3:
    q.requires_grad_(True)
tot_uni.backward()
for p, q in zip(model3.parameters(), ref3.parameters()):
    assert torch.allclose(p.grad, q.grad, atol=1e-6), (p.grad - q.grad).abs().max()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_grad_bucket_reducer_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_reducer_single_process_is_a_noop_average():
    from mdvit_amd.parallel import GradBucketReducer
    lin = torch.nn.Linear(4, 4)
    red = GradBucketReducer(lin.parameters())
    red.zero_grad()
    red.arm()
    lin(torch.ones(2, 4)).sum().backward()
    red.finish()
    assert torch.allclose(lin.weight.grad, torch.full((4, 4), 2.0))


def test_the_256_tile_takes_the_shapes_that_fill_whole_rounds_of_the_chip():
    """mdvit_gemm_ph_prefers (host logic, csrc/gemm_ph.hip): the share of real output in the chip's rounds of 256 x 256 tiles decides; K must hold two K tiles of
    the mode (32 k for two planes, 64 k for one); the A/B hook switches it off / forces it."""
    from mdvit_amd import _lib
    lib = _lib.load()
    f = lib.mdvit_gemm_ph_prefers
    assert f(32768, 960, 320, 2) == 1 and f(8192, 2048, 512, 2) == 1 and f(8192, 1536, 512, 2) == 1 and f(32768, 1280, 320, 2) == 1
    assert f(16384, 1280, 320, 2) == 0 and f(4096, 2048, 512, 2) == 0 and f(32768, 320, 1280, 2) == 0 and f(8192, 512, 2048, 2) == 0
    assert f(32768, 512, 2048, 2) == 1                      # the stage-3 shapes of the 128-image step
    assert f(32768, 960, 48, 2) == 0 and f(32768, 960, 64, 2) == 1 and f(32768, 960, 64, 1) == 0 and f(32768, 960, 128, 1) == 1 and f(32768, 962, 320, 2) == 0
    try:
        lib.mdvit_gemm_ph_config(-1)
        assert f(32768, 960, 320, 2) == 0
        lib.mdvit_gemm_ph_config(1)
        assert f(4096, 2048, 512, 2) == 1 and f(4096, 2048, 32, 2) == 0
    finally:
        lib.mdvit_gemm_ph_config(0)


def test_sweep_graph_audit_reports_engine_launched_work_without_gpu():
    """ops.audit_sweep_graph walks an autograd graph and names what the ENGINE would launch in a backward from the root -- torch-native nodes with a kernel in their
    backward, tensors that receive more than one gradient -- i.e. the work that stays on the forward's stream when a sweep is moved to a stream of its own
    (train._aux_graph_is_ours moves the aux sweep only when both lists are empty and caches the verdict on the model)."""
    from mdvit_amd import ops, train
    x = torch.randn(5, 3, requires_grad=True)
    w = torch.randn(3, 3, requires_grad=True)
    h = x @ w                                   # MmBackward0: a kernel in its backward
    clean = (h.view(15).view(5, 3)).t()         # views only on top of it
    native, fanin = ops.audit_sweep_graph(clean)
    assert native == ["MmBackward0"] and fanin == []
    two = h.view(15) + h.t().reshape(15)        # h has two consumers: autograd adds the two gradients itself
    native, fanin = ops.audit_sweep_graph(two)
    assert "MmBackward0" in native and any(n.startswith("MmBackward0") for n in fanin)
    leaf_twice = x.view(15) + x.t().reshape(15)           # a leaf used twice: AccumulateGrad fan-in is the caller's business (sunk weights hand None)
    assert ops.audit_sweep_graph(leaf_twice)[1] == []

    class M(torch.nn.Module):
        pass
    m = M()
    assert train._aux_graph_is_ours(m, two) is False and m._aux_sweep_graph_ok is False and m._aux_sweep_graph_findings[1]
    assert train._aux_graph_is_ours(m, clean.detach().requires_grad_(True) * 1.0) is False      # cached verdict: the graph is audited once per model
    m2 = M()
    assert train._aux_graph_is_ours(m2, x.view(15).view(3, 5)) is True
