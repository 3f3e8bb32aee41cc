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
    assert f is not None and ("vine_step_quad_kernel<0, true, 0>" in d["kernel_stats"]["name"]
                              or "vine_step_quad_kernel<0, true, 0, false>" in d["kernel_stats"]["name"])
    for inst in ("vine_step_quad_kernel<0, true, 1>", "vine_step_quad_kernel<0, true, 2>"):
        f2, d2 = bench._pmc_summary(inst)
        # (round-5 summaries name the four-argument template: the plain step is its `false` instantiation)
        assert f2 is not None and f2 != f and (inst in d2["kernel_stats"]["name"] or inst[:-1] + ", false>" in d2["kernel_stats"]["name"])
    f3, d3 = bench._pmc_summary("vine_step_kernel<0, true, 0>")
    assert f3 is not None and "vine_step_kernel<0, true, 0>" in d3["kernel_stats"]["name"]
    # the default line's compute roofline reproduces from the summary by hand
    per_wave = d["SQ_INSTS_VALU"]["mean_per_launch"] / d["SQ_WAVES"]["mean_per_launch"]
    ms = d["kernel_stats"]["avg_ns"] * 1e-6
    c = bench.compute_roofline("vine_step_quad_kernel", 16384, ms, "vine_step_quad_kernel<0, true, 0>")
    assert abs(c["valu_issue_frac"] - per_wave * 1024 / (ms * 1e-3) / (1024 * 2.4e9 / 2)) < 1e-12
    assert 0.15 < c["valu_issue_frac"] < 0.3 and abs(c["flop_frac"] - 27149.0 * 16384 / (ms * 1e-3) / 157.3e12) < 1e-12
    assert json.dumps(c)          # serialisable


def test_traffic_summary_and_on_path_rooflines_reproduce_by_hand(tmp_path, monkeypatch, capsys):
    """scripts/pmc_traffic_summary.py on a hand-made pair of counter passes + a kernel-stats table (the x2 FETCH_SIZE
    correction, per-dispatch sums over the XCD rows, the in-situ duration from the graphed trace), and
    bench_support.ppo_path_rooflines on the newest COMMITTED summary: every figure of a row follows from the file it names
    and from the shapes of the agent."""
    import types
    root = tmp_path / "ppo_traffic_x"
    for sub in ("pmc_fetch/a", "pmc_write/a", "trace/a"):
        (root / sub).mkdir(parents=True)
    hdr = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    k = "(anonymous namespace)::trunk_phases_kernel((anonymous namespace)::TrunkPhasesArgs)"
    k2 = "void (anonymous namespace)::wgrad_cat_wide_kernel<6, 2>(int, int)"
    (root / "pmc_fetch/a/1_counter_collection.csv").write_text(
        hdr + '1,"%s",FETCH_SIZE,100\n1,"%s",FETCH_SIZE,50\n2,"%s",FETCH_SIZE,170\n3,"%s",FETCH_SIZE,10\n' % (k, k, k, k2))
    (root / "pmc_write/a/1_counter_collection.csv").write_text(
        hdr + '1,"%s",WRITE_SIZE,200\n2,"%s",WRITE_SIZE,220\n3,"%s",WRITE_SIZE,40\n' % (k, k, k2))
    (root / "trace/a/1_kernel_stats.csv").write_text(
        'Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n"%s",4,600000,150000,75.0,1,2,0\n"%s",4,200000,50000,25.0,1,2,0\n' % (k, k2))
    mod = _load(os.path.join(REPO, "scripts", "pmc_traffic_summary.py"), "pmc_traffic_summary")
    monkeypatch.setattr(sys, "argv", ["pmc_traffic_summary.py", str(root)])
    mod.main()
    out = json.load(open(root / "pmc_summary.json"))
    t = out["trunk_phases_kernel"]
    assert t["pmc_launches"] == 2 and t["FETCH_SIZE_KiB"] == 160.0 and t["WRITE_SIZE_KiB"] == 210.0      # (100 + 50 | 170) / 2
    assert t["traffic_bytes"] == (2 * 160.0 + 210.0) * 1024 and t["avg_ns"] == 150000.0
    assert abs(t["hbm_frac_of_8TBs"] - t["traffic_bytes"] / 150000.0 / 8000.0) < 1e-15
    assert out["wgrad_cat_wide_kernel<6, 2>"]["traffic_bytes"] == (2 * 10 + 40) * 1024
    capsys.readouterr()
    # ---- the bench rows, from the committed profile
    sys.path.insert(0, REPO)
    from vine_robot_isaacgymenvs_amd.learning import bench_support
    net = types.SimpleNamespace(rnn_units=256, units=[256, 128, 64])
    agent = types.SimpleNamespace(model=types.SimpleNamespace(a2c_network=net), minibatch_size=32768, seq_len=4, obs_shape=(28,),
                                  actions_num=2, mini_epochs_num=4, num_minibatches=8, num_actors=16384, horizon_length=16,
                                  _fast={"f32_split": 6})
    rows = bench_support.ppo_path_rooflines(agent)
    assert rows, "no committed profiles/r*/ppo_traffic_*_pmc_summary.json"
    by = {r["kernel"].split("<")[0]: r for r in rows}
    prof = json.load(open(os.path.join(REPO, rows[0]["source"])))
    tr = by["trunk_phases_kernel"]
    assert tr["launches_per_iteration"] == 32 and tr["in_situ_us"] == prof["trunk_phases_kernel"]["avg_ns"] / 1e3
    # the LSTM forward phase's bytes by hand: x + h0, c0 + weights + done flags in; h [B, T + 1, H], c, gates out (B = 8192, T = 4)
    B, T, H, wpad = 8192, 4, 256, 96
    fwd = B * T * wpad * 2 + 2 * B * H * 4 + 4 * H * (wpad + H) * 2 + B * T + B * (T + 1) * H * 2 + ((T - 1) * B * H * 2 + B * H * 4) + T * B * 4 * H * 2
    assert tr["phases"]["lstm_forward"] == fwd == 132874240
    assert tr["algorithmic_bytes"] == sum(tr["phases"].values())
    assert abs(tr["hbm_frac"] - tr["algorithmic_bytes"] / tr["in_situ_us"] / 1e3 / 8000.0) < 1e-12
    assert tr["traffic_bytes"] == prof["trunk_phases_kernel"]["traffic_bytes"]
    assert 0.9 < tr["traffic_over_algorithmic"] < 1.3                  # the counters agree with the byte formulas
    assert by["vine_step_quad_kernel"]["algorithmic_bytes"] == 320 * 16384 and by["vine_step_quad_kernel"]["bound"] == "valu"
    agent.rollout_step_launches = 3                                    # the one-launch rollout step: the ROLL instantiation's row
    roll = {r["kernel"].split("<")[0]: r for r in bench_support.ppo_path_rooflines(agent)}["vine_step_quad_kernel"]
    assert roll["kernel"].endswith(", true>") and roll["algorithmic_bytes"] == 320 * 16384 + 16384 * (256 * 4 + 9 * 4 + 1)
    assert any(r.get("bound") == "mfma" for r in rows) and json.dumps(rows)
