"""Test-side driver of the PRODUCT library (libvine_hip.so) through its C ABI, with torch device buffers.
Mirrors oracle.vine_oracle.OracleEnv so parity tests can treat both alike."""
import ctypes as C

import numpy as np
import torch

from vine_robot_isaacgymenvs_amd import abi, native


class HipEnv:
    kernel = None        # "lane" / "quad": force the one-lane or the four-lanes-per-env step kernel (None: by size)

    def __init__(self, cfg, device_id=0):
        import os
        self.lib = native.load()
        # the parity tests compare every field, also those the step only stores on request: created with introspection ON
        # (a private copy of the config; switching it on later would make the first step refresh the body states)
        cfg = type(cfg).from_buffer_copy(cfg)
        cfg.set_flag(abi.FLAG_INTROSPECT, True)
        self.cfg = cfg
        self.n = cfg.num_envs
        self.dev = torch.device("cuda", device_id)
        self.num_obs = self.lib.vine_num_obs(C.byref(cfg))
        self.state_t = torch.zeros((abi.VF_COUNT, self.n), device=self.dev, dtype=torch.float32)
        h = C.c_void_p()
        old = os.environ.get("VINE_STEP_KERNEL")
        if self.kernel:
            os.environ["VINE_STEP_KERNEL"] = self.kernel      # read by vine_create
        try:
            native.check(self.lib.vine_create(C.byref(cfg), device_id, self.state_t.data_ptr(), C.byref(h)), self.lib)
        finally:
            if self.kernel:
                if old is None:
                    os.environ.pop("VINE_STEP_KERNEL", None)
                else:
                    os.environ["VINE_STEP_KERNEL"] = old
        self.h = h
        self.obs_t = torch.zeros((self.n, self.num_obs), device=self.dev)
        self.rew_t = torch.zeros(self.n, device=self.dev)
        self.reset_t = torch.ones(self.n, device=self.dev, dtype=torch.long)
        self.progress_t = torch.zeros(self.n, device=self.dev, dtype=torch.long)
        self.timeouts_t = torch.zeros(self.n, device=self.dev, dtype=torch.bool)
        self.reward_matrix_t = None
        self._rv = None

    def set_introspection(self, on):
        native.check(self.lib.vine_set_introspection(self.h, int(bool(on))), self.lib)

    def stats(self, index_to_view=0):
        out = torch.zeros(abi.NUM_STATS, device=self.dev)
        native.check(self.lib.vine_stats(self.h, self.rew_t.data_ptr(), self.progress_t.data_ptr(), int(index_to_view),
                                         out.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream), self.lib)
        torch.cuda.synchronize(self.dev)
        return out.cpu().numpy()

    # numpy views of the device buffers (copies)
    @property
    def state(self):
        return self.state_t.cpu().numpy().astype(np.float64)

    def set_state(self, st):
        self.state_t.copy_(torch.as_tensor(np.asarray(st), dtype=torch.float32))

    @property
    def reset_buf(self):
        return self.reset_t.cpu().numpy()

    @property
    def progress(self):
        return self.progress_t.cpu().numpy()

    def set_flags(self, reset, progress):
        self.reset_t.copy_(torch.as_tensor(np.asarray(reset), dtype=torch.long))
        self.progress_t.copy_(torch.as_tensor(np.asarray(progress), dtype=torch.long))

    def bind_reward_matrix(self):
        self.reward_matrix_t = torch.zeros((self.n, abi.NUM_REWARDS), device=self.dev)
        native.check(self.lib.vine_bind_reward_matrix(self.h, self.reward_matrix_t.data_ptr()), self.lib)

    def bind_reset_values(self, values):
        if values is None:
            self._rv = None
            native.check(self.lib.vine_bind_reset_values(self.h, None), self.lib)
        else:
            self._rv = torch.as_tensor(np.asarray(values, np.float32).reshape(self.n, 10)).to(self.dev).contiguous()
            native.check(self.lib.vine_bind_reset_values(self.h, self._rv.data_ptr()), self.lib)

    @property
    def step_count(self):
        return self.lib.vine_get_step_count(self.h)

    @step_count.setter
    def step_count(self, v):
        native.check(self.lib.vine_set_step_count(self.h, int(v)), self.lib)

    def step(self, actions, sync=True):
        a = torch.as_tensor(np.asarray(actions, np.float32).reshape(self.n, 2)).to(self.dev).contiguous()
        return self.step_t(a, sync)

    def step_t(self, a, sync=True):
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        native.check(self.lib.vine_step(self.h, a.data_ptr(), self.obs_t.data_ptr(), self.rew_t.data_ptr(),
                                        self.reset_t.data_ptr(), self.progress_t.data_ptr(), self.timeouts_t.data_ptr(),
                                        stream), self.lib)
        if not sync:
            return None
        torch.cuda.synchronize(self.dev)
        return (self.obs_t.cpu().numpy(), self.rew_t.cpu().numpy(), self.reset_t.cpu().numpy(),
                self.timeouts_t.cpu().numpy().astype(np.uint8))

    def reset_idx(self, env_ids):
        ids = torch.as_tensor(np.asarray(env_ids, np.int64)).to(self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        native.check(self.lib.vine_reset_idx(self.h, ids.data_ptr(), ids.numel(), self.rew_t.data_ptr(),
                                             self.reset_t.data_ptr(), self.progress_t.data_ptr(), stream), self.lib)
        torch.cuda.synchronize(self.dev)

    def close(self):
        if self.h:
            torch.cuda.synchronize(self.dev)
            self.lib.vine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
