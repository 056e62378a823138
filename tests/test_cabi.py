"""The C-ABI library loads and exports every symbol include/ohmhip.h declares (no compute calls: CPU only)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ohmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ohmhip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ohm_amd
    lib = C.CDLL(ohm_amd.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"libohmhip.so does not export: {missing}"
    # and the Python binding covers all of them
    unbound = [s for s in declared if s not in ohm_amd.EXPORTED_SYMBOLS]
    assert not unbound, f"ohm_amd._lib does not bind: {unbound}"


def test_error_strings_and_defaults():
    from ohm_amd import _lib as L
    assert L.lib.ohmhip_error_string(0) == b"ok"
    assert b"capacity" in L.lib.ohmhip_error_string(L.ERR_CAPACITY)
    cfg = L.MapConfig()
    L.lib.ohmhip_map_config_default(C.byref(cfg))
    assert cfg.resolution == pytest.approx(0.1) and list(cfg.region_dim) == [32, 32, 32]
    # same constants as the CPU oracle / reference defaults (ohm/OccupancyMap.cpp:205-213)
    from oracle.oracle import OracleMap
    om = OracleMap()
    assert cfg.hit_value == om.hit_value() and cfg.miss_value == om.miss_value()
    assert cfg.min_value == -2.0 and abs(cfg.max_value - 3.511) < 1e-6
    assert [L.lib.ohmhip_layer_voxel_bytes(i) for i in range(9)] == [4, 8, 24, 4, 4, 4, 8, 8, 8]


def test_no_device_is_reported_not_faked():
    """Without a GPU the product path must refuse to run (no CPU fallback anywhere)."""
    import ohm_amd
    if ohm_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ohm_amd.OhmHipError):
        ohm_amd.GpuMap(ohm_amd.OccupancyMap())


def test_product_path_never_imports_oracle():
    """Nothing under ohm_amd/ or include/ may reference the oracle."""
    bad = []
    for base in ("ohm_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"import\s+oracle|from\s+oracle|ohm_oracle|libohm_oracle", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_python_constants_match_host_libm():
    import numpy as np
    import ohm_amd
    from oracle import oracle as O
    for p in (0.9, 0.45, 0.5, 0.2, 0.55, 0.7):
        assert ohm_amd.probability_to_value(p) == np.float32(O.lib.oracle_probability_to_value(p))


def test_null_and_out_of_range_arguments_are_rejected():
    """Entry points that need no device validate their arguments the same way with or without a GPU."""
    import numpy as np
    from ohm_amd import _lib as L
    invalid = L.ERR_INVALID_ARG
    assert L.lib.ohmhip_map_set_batch_coalescing(None, 4096) == invalid
    assert L.lib.ohmhip_map_set_async_launch(None, 1) == invalid
    assert L.lib.ohmhip_map_set_region_ownership(None, 2, 0, 0) == invalid
    assert L.lib.ohmhip_map_integrate_rays(None, None, 0, None, None, 0, None) == invalid
    assert L.lib.ohmhip_map_sync(None) == invalid
    keys = np.zeros((4, 3), dtype=np.int16)
    owners = np.zeros(4, dtype=np.uint32)
    assert L.lib.ohmhip_region_owner(keys.ctypes.data, 4, -1, 2, owners.ctypes.data) == invalid
    assert L.lib.ohmhip_region_owner(keys.ctypes.data, 4, 16, 2, owners.ctypes.data) == invalid
    assert L.lib.ohmhip_region_owner(None, 4, 0, 2, owners.ctypes.data) == invalid
    assert L.lib.ohmhip_region_owner(keys.ctypes.data, 4, 0, 2, owners.ctypes.data) == L.OK
    assert L.lib.ohmhip_region_owner(None, 0, 0, 2, None) == L.OK


def test_core_abi_list_is_what_the_binding_core_calls():
    """include/ohmhip.h names the (at most 20) entry points a reference-side binding needs: exactly the ohmhip_* calls of
    the compiled binding core (ohm_amd/host/ref_adaptor/private/HipBindingCore.cpp) plus ohmhip_error_string for the glue's
    log lines; tuning / measurement knobs are tagged OHMHIP_EXPERIMENTAL and are never among them."""
    header = open(os.path.join(ROOT, "include", "ohmhip.h")).read()
    core = sorted(set(sum((ln.split(":", 1)[1].split() for ln in header.splitlines() if "OHMHIP_CORE_ABI:" in ln), [])))
    assert 10 <= len(core) <= 20, core
    src = open(os.path.join(ROOT, "ohm_amd", "host", "ref_adaptor", "private", "HipBindingCore.cpp")).read()
    called = sorted(set(re.findall(r"\b(ohmhip_[a-z_0-9]+)\s*\(", src)))
    assert sorted(set(called) | {"ohmhip_error_string"}) == core
    experimental = set(re.findall(r"OHMHIP_EXPERIMENTAL\s+int\s+(ohmhip_[a-z_0-9]+)", header))
    assert len(experimental) >= 6 and not (experimental & set(core))
    assert set(core) <= set(_declared_symbols())
