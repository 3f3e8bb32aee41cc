"""Stand-alone timing of the persistent LSTM sequence kernels at the update's shape (8192 sequences x 4 steps, H 256,
[x (96) | h] operand), rotating over several operand/output sets so that every launch runs on cold caches.
Knob from the environment (read once per process by the library): VINE_SEQ_ABLATE (bit 0: no global stores,
bit 1: no weight reloads); VINE_HIP_LIB selects an A/B build (scripts/ab_build.sh).  `python scripts/ubench/lstm_seq_bench.py sweep`
runs a list of settings, each in a child process."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def one():
    import torch
    from vine_robot_isaacgymenvs_amd.learning import fused
    dev = torch.device("cuda:0")
    lib = fused._lib()
    B, T, H, width, wpad, SETS = 8192, int(os.environ.get("SEQ_T", "4")), 256, 92, 96, 4
    bf = fused.lp_dtype()
    torch.manual_seed(0)
    w_ih = (torch.randn(4 * H, width, device=dev) / 10).to(bf)
    w_hh = (torch.randn(4 * H, H, device=dev) / 16).to(bf)
    wtile = torch.empty(4 * H * (wpad + H), device=dev, dtype=bf)
    whh_tiled = torch.empty(4 * H * H, device=dev, dtype=bf)
    prep = fused.CopyBatch()
    prep.add_lstm_tiles(w_ih, w_hh, wpad, wtile, whh_tiled)
    prep.flush(wtile)
    bias = torch.randn(4 * H, device=dev) * 0.1
    sets = []
    for _ in range(SETS):
        x = torch.zeros(B * T, wpad, device=dev, dtype=bf)
        x[:, :width] = (torch.randn(B * T, width, device=dev) * 0.7).to(bf)
        h0, c0 = torch.randn(B, H, device=dev) * 0.5, torch.randn(B, H, device=dev) * 0.5
        dones = (torch.rand(B * T, device=dev) < 0.25).to(torch.uint8)
        bufs = fused._lstm_state_buffers(x, w_hh, h0, c0, dones, T, True)
        g_out = torch.randn(B * T, H, device=dev) * 0.1
        sets.append((x, h0, c0, dones, bufs, g_out))
    st = torch.cuda.current_stream().cuda_stream

    def fwd(i):
        x, h0, c0, dones, (out, c_all, gates, hp), g_out = sets[i % SETS]
        rc = lib.vine_lstm_seq_forward_mfma(B, T, H, wpad, x.data_ptr(), wpad, hp.data_ptr(), T * H, wtile.data_ptr(),
                                            bias.data_ptr(), c0.data_ptr(), dones.data_ptr(), out.data_ptr(), c_all.data_ptr(),
                                            gates.data_ptr(), 0, None, None, st)
        assert rc == 0

    dG = [torch.empty(B * T, 4 * H, device=dev, dtype=bf) for _ in range(SETS)]
    part = torch.empty(B // 32, 4 * H, device=dev)

    def bwd(i):
        x, h0, c0, dones, (out, c_all, gates, hp), g_out = sets[i % SETS]
        rc = lib.vine_lstm_seq_backward_mfma(B, T, H, g_out.data_ptr(), whh_tiled.data_ptr(), gates.data_ptr(), c_all.data_ptr(),
                                             c0.data_ptr(), dones.data_ptr(), dG[i % SETS].data_ptr(), part.data_ptr(), 0, None, 0, st)
        assert rc == 0

    res = {}
    for name, f in (("fwd", fwd), ("bwd", bwd)):
        for i in range(SETS * 2):
            f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 40
        for i in range(n):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n * 1e3
    print("T=%d ABLATE=%s  fwd %.1f us  bwd %.1f us" % (T, os.environ.get("VINE_SEQ_ABLATE", "0"), res["fwd"], res["bwd"]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        settings = sys.argv[2:] or ["-", "A1", "A2", "A3"]
        for s in settings:
            env = dict(os.environ)
            env.pop("VINE_SEQ_ABLATE", None)
            if s.startswith("A"):
                env["VINE_SEQ_ABLATE"] = s[1:]
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
    else:
        one()
