"""The TensorBoard event-file writer (utils/tfevents.py): CRC-32C known answers, TFRecord framing, and a round trip of
the scalars the agent logs (rl_games' tag names; the reference logs through tensorboardX, utils/rlgames_utils.py:95-148)."""
import glob
import os
import struct

import pytest

from vine_robot_isaacgymenvs_amd.utils import tfevents


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert tfevents.crc32c(b"") == 0
    assert tfevents.crc32c(b"123456789") == 0xE3069283
    assert tfevents.crc32c(bytes(32)) == 0x8A9136AA
    assert tfevents.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfevents.crc32c(bytes(range(32))) == 0x46DD794E


def test_event_file_round_trip(tmp_path):
    w = tfevents.EventFileWriter(str(tmp_path))
    rows = [("rewards/step", 1.5, 0), ("losses/a_loss", -0.25, 16384 * 16), ("info/kl", 3e-3, 2 ** 40), ("info/epochs", 7.0, -1)]
    for tag, v, step in rows:
        w.add_scalar(tag, v, step)
    w.close()
    files = glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))
    assert len(files) == 1
    got = tfevents.read_scalars(files[0])
    assert [(t, s) for t, _v, s, _w in got] == [(t, s) for t, _v, s in rows]
    for (_t, v, _s, _w), (_t2, v2, _s2) in zip(got, rows):
        assert v == struct.unpack("<f", struct.pack("<f", v2))[0]
    # first record: the version header TensorBoard looks for; framing = length, crc(length), bytes, crc(bytes)
    data = open(files[0], "rb").read()
    (n,) = struct.unpack("<Q", data[:8])
    assert b"brain.Event:2" in data[12:12 + n]
    corrupt = bytearray(data)
    corrupt[20] ^= 1
    bad = os.path.join(str(tmp_path), "bad")
    open(bad, "wb").write(bytes(corrupt))
    with pytest.raises(ValueError):
        tfevents.read_scalars(bad)


def test_scalar_log_writes_csv_and_events(tmp_path):
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import ScalarLog
    log = ScalarLog(str(tmp_path))
    log.add_scalar("performance/step_inference_rl_update_fps", 2.37e7, 262144)
    log.flush()
    assert "performance/step_inference_rl_update_fps,23700000.0,262144" in open(log.path).read()
    got = tfevents.read_scalars(log.events.path)
    assert got[0][0] == "performance/step_inference_rl_update_fps" and got[0][2] == 262144 and abs(got[0][1] - 2.37e7) < 2.0


def test_scalars_beyond_float32_become_inf(tmp_path):
    """A diverged loss held as a Python float / float64 (|v| > 3.4e38) must not abort the run: it is logged as
    +-inf, as tensorboardX would; NaN stays NaN."""
    w = tfevents.EventFileWriter(str(tmp_path))
    for i, v in enumerate([1e39, -1e39, float("inf"), float("nan"), 3.0e38]):
        w.add_scalar("losses/c_loss", v, i)
    w.close()
    got = [v for _t, v, _s, _w in tfevents.read_scalars(glob.glob(os.path.join(str(tmp_path), "events.out.tfevents.*"))[0])]
    assert got[0] == float("inf") and got[1] == float("-inf") and got[2] == float("inf") and got[3] != got[3]
    assert got[4] == struct.unpack("<f", struct.pack("<f", 3.0e38))[0]
