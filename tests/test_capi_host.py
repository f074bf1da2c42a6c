"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ge2e_hip.h declares, its parameter table is the reference's state_dict key set, the host-side
dropout stream equals the oracle's, and the product path refuses to run without the GPU (no fallback).
No compute call is made here."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as O
from speaker_embedding_torch_amd import _build, _lib
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    _build.build()          # hipcc cross-compiles for gfx950 without a GPU (no-op when up to date)


def make_hp(layers=3, p=0.1):
    from argparse import Namespace
    return Namespace(Sound=Namespace(Mel_Dim=80),
                     GE2E=Namespace(Embedding_Size=256,
                                    Positional_Encoding=Namespace(Max_Position=1024, Dropout_Rate=p),
                                    Transformer=Namespace(Num_Layers=layers, Head=4, Dropout_Rate=p)))


def test_header_symbols_all_exported():
    text = open(os.path.join(REPO, "include", "ge2e_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ge2e_[a-z0-9_]+)\s*\(", text))
    declared.discard("ge2e_bucket_cb")
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ge2e_abi_version() == _lib.ABI_VERSION


def test_param_table_is_reference_state_dict_order():
    h = _lib.Handle()
    specs = O.param_specs()
    assert h.param_names == [n for n, _ in specs]
    assert h.param_numel == [int(np.prod(s)) for _, s in specs]
    assert h.param_offset == list(np.cumsum([0] + h.param_numel[:-1]))
    assert h.param_total == 2456321


def test_create_rejects_unsupported_configs():
    lib = _lib.load()
    import ctypes as C
    for kw in (dict(emb=128, heads=2), dict(heads=8), dict(layers=0), dict(layers=9), dict(ffn=1000), dict(precision=7),
               dict(tf_dropout=1.0)):
        base = dict(mel_dim=80, emb=256, heads=4, layers=3, ffn=1024, max_position=1024, pe_dropout=0.1,
                    tf_dropout=0.1, ln_eps=1e-5, precision=0)
        base.update(kw)
        cfg = _lib.Config(*[base[k] for k, _ in _lib.Config._fields_])
        out = C.c_void_p()
        assert lib.ge2e_create(C.byref(cfg), C.byref(out)) < 0, kw
        assert not out.value
    assert lib.ge2e_create(None, None) < 0


def test_workspace_sizes_and_limits():
    h32, h16 = _lib.Handle(precision=_lib.PREC_F32), _lib.Handle(precision=_lib.PREC_BF16)
    for n, t in ((20, 160), (960, 160), (1280, 64), (2560, 180)):
        tr, ev = h32.workspace_bytes(n, t, True), h32.workspace_bytes(n, t, False)
        assert tr > ev > 0
        assert h16.workspace_bytes(n, t, True) < tr
    assert h32.workspace_bytes(0, 160, True) == 0
    assert h32.max_frames() == 1024                      # = Max_Position: whatever reference Modules.py:107-109 accepts
    assert _lib.Handle(max_position=400).max_frames() == 400
    assert h32.workspace_bytes(4, 512, True) > h32.workspace_bytes(4, 288, True)
    # config 5 (2560 x 180, bf16 storage) must fit one 288 GB MI355X many times over
    assert h16.workspace_bytes(2560, 180, True) < 32 * 2 ** 30


def test_dropout_stream_matches_oracle():
    lib = _lib.load()
    for seed, step, site in ((0, 0, 0), (1234, 7, 5), (2 ** 40 + 3, 99999, 12)):
        key = lib.ge2e_drop_key(seed, step, site)
        assert key == O.drop_key(seed, step, site)
        ref = O.drop_keep(key, 4096, 0.1, start=10 ** 6)
        got = np.array([lib.ge2e_drop_keep(key, 10 ** 6 + i, 0.1) for i in range(4096)], bool)
        assert (ref == got).all()


def test_module_state_dict_is_reference_key_set():
    m = GE2E(make_hp())
    sd = m.state_dict()
    expected = [n for n, _ in O.param_specs()]
    expected.insert(3, "positional_encoding.pe")      # the buffer sits after alpha's module params
    assert sorted(sd.keys()) == sorted(expected) and len(sd) == 44
    for name, shape in O.param_specs():
        assert tuple(sd[name].shape) == shape, name
    assert tuple(sd["positional_encoding.pe"].shape) == (1, 256, 1024)
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in O.param_specs()]
    # pe is the reference's sinusoid table (Modules.py:84-90)
    # (fp32 sin/cos of arguments up to 1023 differ by a few ulp of the ARGUMENT between torch and numpy)
    assert np.abs(sd["positional_encoding.pe"][0].numpy().T - O.sinusoid_pe(1024, 256)).max() < 2e-4
    assert np.abs(sd["positional_encoding.pe"][0].numpy().T[:300] - O.sinusoid_pe(1024, 256)[:300]).max() < 5e-5
    # strict load of a reference-shaped checkpoint dict (Train.py:285)
    ref_like = {k: torch.from_numpy(v.copy()) for k, v in O.formula_params().items()}
    ref_like["positional_encoding.pe"] = sd["positional_encoding.pe"].clone()
    m.load_state_dict(ref_like, strict=True)
    # reference quirk: TransformerEncoder deep-copies one layer -> identical layers at init
    m2 = GE2E(make_hp())
    a, b = m2.transformer.layers[0].state_dict(), m2.transformer.layers[2].state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert float(m2.positional_encoding.alpha) == 1.0 and float(m2.prenet.bias.abs().sum()) == 0.0


def test_no_cpu_fallback():
    m = GE2E(make_hp())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(4, 80, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GE2E_Loss()(torch.zeros(4, 256), 2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB", str(tmp_path / "libge2e_hip.so"))
    with pytest.raises(RuntimeError, match="has not been built"):
        _lib.load()


def test_product_code_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import, load or execute it."""
    pkg = os.path.join(REPO, "speaker_embedding_torch_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle|import_module\(.*oracle|#include.*oracle", re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                assert not pat.search(open(os.path.join(root, f)).read()), f


def test_build_tracks_every_source_and_loader_refuses_a_stale_binary(monkeypatch):
    """the dependency list is csrc/*.hip + csrc/*.cuh + the header (no hand-kept list), and a binary whose
    embedded source hash differs from the csrc/ beside it must not load."""
    import glob
    from speaker_embedding_torch_amd import _build
    deps = set(_build.dependencies())
    csrc = os.path.join(REPO, "speaker_embedding_torch_amd", "csrc")
    assert set(glob.glob(os.path.join(csrc, "*.cuh")) + glob.glob(os.path.join(csrc, "*.hip"))) <= deps
    assert os.path.join(REPO, "include", "ge2e_hip.h") in deps
    lib = _lib.load()
    assert lib.ge2e_source_hash().decode() == _build.source_hash() == _build.built_hash()
    assert not _build.needs_build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_build, "source_hash", lambda: "0" * 32)
    assert _build.needs_build()
    with pytest.raises(RuntimeError, match="stale"):
        _lib.load()


def test_precision_a_hyper_parameter_file_asks_for():
    """Use_Mixed_Precision means float16 + GradScaler in the reference (Train.py:134,145); bf16 is this build's optional key."""
    from argparse import Namespace
    from speaker_embedding_torch_amd.Modules import default_precision
    assert default_precision(Namespace(Use_Mixed_Precision=False)) == "fp32"
    assert default_precision(Namespace(Use_Mixed_Precision=True)) == "fp16"
    assert default_precision(Namespace(Use_Mixed_Precision=True, Mixed_Precision_Dtype="bf16")) == "bf16"
    assert default_precision(Namespace(Use_Mixed_Precision=False, Mixed_Precision_Dtype="bf16")) == "fp32"
    from speaker_embedding_torch_amd.Arg_Parser import Load_Hyper_Parameters
    hp = Load_Hyper_Parameters(os.path.join(REPO, "speaker_embedding_torch_amd", "Hyper_Parameters.yaml"))
    assert default_precision(hp) == "bf16"          # the shipped recipe = BASELINE.json configs[1]


def test_options_are_an_api_not_the_environment():
    """the development switches are ge2e_set_option entries with compiled-in defaults; no product source reads the
    environment for them, and the loader forwards GE2E_<NAME> variables only under GE2E_DEV_SWITCHES=1."""
    lib = _lib.load()
    names = _lib.option_names()
    assert "no_overlap" in names and "debug_bwd_stop" in names and len(names) == len(set(names))
    header = open(os.path.join(REPO, "include", "ge2e_hip.h")).read()
    for n in names:
        assert f'"{n}"' in header, f"option {n} is not documented in include/ge2e_hip.h"
    assert _lib.get_option("no_ws_gemm") == 0 and _lib.get_option("debug_bwd_stop") == -1
    _lib.set_option("no_ws_gemm", 1)
    try:
        assert _lib.get_option("no_ws_gemm") == 1       # read at every call: nothing latched
    finally:
        _lib.set_option("no_ws_gemm", 0)
    with pytest.raises(KeyError):
        _lib.set_option("no_such_option", 1)
    assert lib.ge2e_set_option(None, 1) != 0
    csrc = os.path.join(REPO, "speaker_embedding_torch_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f} reads the environment"
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom speaker_embedding_torch_amd import _lib\nprint(_lib.get_option('no_kl_gemm'))" % REPO)
    plain = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GE2E_NO_KL_GEMM="1"), capture_output=True, text=True, timeout=300)
    dev = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GE2E_NO_KL_GEMM="1", GE2E_DEV_SWITCHES="1"), capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and plain.stdout.strip() == "0", plain.stderr[-500:]
    assert dev.returncode == 0 and dev.stdout.strip() == "1", dev.stderr[-500:]
