"""Host-side tooling around the kernels: the DPP hazard checker of the four-lane step kernel and bench.py's matching of
committed PMC summaries to kernel instantiations (CPU only)."""
import importlib.util
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dpp_hazard_checker_flags_an_unprotected_read(tmp_path, monkeypatch, capsys):
    """scripts/check_dpp_hazards.py on hand-written assembly: a v_fmac_f32_dpp that reads a VGPR written by the previous
    VALU instruction is reported; the same read two instructions later, or behind `s_nop 1`, is not."""
    chk = _load(os.path.join(REPO, "scripts", "check_dpp_hazards.py"), "check_dpp_hazards")
    dpp = "quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
    bad = ["_Zkernel_a:", "\tv_mul_f32_e32 v5, v1, v2", "\tv_fmac_f32_dpp v7, v5, v3 " + dpp]
    ok1 = ["_Zkernel_b:", "\tv_mul_f32_e32 v5, v1, v2", "\tv_add_f32_e32 v8, v1, v2", "\tv_add_f32_e32 v9, v1, v2",
           "\tv_fmac_f32_dpp v7, v5, v3 " + dpp]
    ok2 = ["_Zkernel_c:", "\tv_mul_f32_e32 v5, v1, v2", "\ts_nop 1", "\tv_fmac_f32_dpp v7, v5, v3 " + dpp]
    one = ["_Zkernel_d:", "\tv_mul_f32_e32 v5, v1, v2", "\tv_add_f32_e32 v8, v1, v2", "\tv_mul_f32_dpp v7, v5, v3 " + dpp]
    f = tmp_path / "t.s"
    f.write_text("\n".join(bad + ok1 + ok2 + one) + "\n")
    monkeypatch.setattr(sys, "argv", ["check_dpp_hazards.py", str(f)])
    assert chk.main() == 1
    out = capsys.readouterr().out
    assert "4 DPP instructions checked, 2 unprotected reads" in out
    assert "_Zkernel_a" in out and "_Zkernel_d" in out and "_Zkernel_b" not in out and "_Zkernel_c" not in out


def test_bench_matches_pmc_summaries_by_instantiation():
    """bench.py reads HBM traffic and VALU counts from the committed PMC summary of the kernel INSTANTIATION that ran
    (template arguments included): the free-space four-lane kernel must not pick up an obstacle kernel's counters."""
    sys.path.insert(0, REPO)
    import bench
    f, d = bench._pmc_summary("vine_step_quad_kernel<0, true, 0>")
    assert f is not None and "vine_step_quad_kernel<0, true, 0>" in d["kernel_stats"]["name"]
    for inst in ("vine_step_quad_kernel<0, true, 1>", "vine_step_quad_kernel<0, true, 2>"):
        f2, d2 = bench._pmc_summary(inst)
        assert f2 is not None and inst in d2["kernel_stats"]["name"] and f2 != f
    f3, d3 = bench._pmc_summary("vine_step_kernel<0, true, 0>")
    assert f3 is not None and "vine_step_kernel<0, true, 0>" in d3["kernel_stats"]["name"]
    # the default line's compute roofline reproduces from the summary by hand
    per_wave = d["SQ_INSTS_VALU"]["mean_per_launch"] / d["SQ_WAVES"]["mean_per_launch"]
    ms = d["kernel_stats"]["avg_ns"] * 1e-6
    c = bench.compute_roofline("vine_step_quad_kernel", 16384, ms, "vine_step_quad_kernel<0, true, 0>")
    assert abs(c["valu_issue_frac"] - per_wave * 1024 / (ms * 1e-3) / (1024 * 2.4e9 / 2)) < 1e-12
    assert 0.15 < c["valu_issue_frac"] < 0.3 and abs(c["flop_frac"] - 27149.0 * 16384 / (ms * 1e-3) / 157.3e12) < 1e-12
    assert json.dumps(c)          # serialisable
