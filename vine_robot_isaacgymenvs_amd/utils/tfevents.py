"""TensorBoard event files without TensorBoard.

rl_games logs through ``tensorboardX.SummaryWriter.add_scalar`` and the reference's observer and task log the same
way (utils/rlgames_utils.py:95-148; the wandb run of V5:618-627 syncs that directory).  Neither package is on the target
image, so this module writes the file format itself: a ``events.out.tfevents.<time>.<host>`` file of TFRecords, each an
``Event`` protocol buffer -- only the four fields a scalar needs are encoded, by hand:

    Event   { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    Summary { repeated Value value = 1; }      Value { string tag = 1; float simple_value = 2; }

TFRecord framing: uint64 length, masked CRC-32C of the length, the bytes, masked CRC-32C of the bytes.  A stock
``tensorboard --logdir runs/<name>/summaries`` (on a machine that has it) shows the curves under rl_games' tag names.
``read_scalars`` parses such a file back (tests; quick looks without TensorBoard)."""
import os
import socket
import struct
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    """CRC-32C (Castagnoli), the checksum of the TFRecord format."""
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1                       # int64 fields: two's complement, ten bytes when negative
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _len_delimited(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _as_f32(value):
    """float -> the nearest float32 as a Python float; finite values beyond float32's range become +-inf (``struct.pack``
    would raise OverflowError and end the training loop over a diverged scalar; tensorboardX logs inf)."""
    value = float(value)
    if value != value or value in (float("inf"), float("-inf")):
        return value
    if abs(value) > 3.4028234663852886e38:
        return float("inf") if value > 0 else float("-inf")
    return value


def _event(wall_time, step=None, file_version=None, tag=None, value=None):
    ev = bytes([(1 << 3) | 1]) + struct.pack("<d", wall_time)
    if step is not None:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _len_delimited(3, file_version.encode())
    if tag is not None:
        val = _len_delimited(1, tag.encode()) + bytes([(2 << 3) | 5]) + struct.pack("<f", _as_f32(value))
        ev += _len_delimited(5, _len_delimited(1, val))
    return ev


class EventFileWriter:
    """``add_scalar(tag, value, step)`` / ``flush()`` / ``close()`` of a TensorBoard SummaryWriter, scalars only."""

    def __init__(self, directory):
        os.makedirs(directory, exist_ok=True)
        name = "events.out.tfevents.%010d.%s.%d" % (int(time.time()), socket.gethostname() or "host", os.getpid())
        self.path = os.path.join(directory, name)
        self._f = open(self.path, "ab")
        self._record(_event(time.time(), file_version="brain.Event:2"))

    def _record(self, data):
        header = struct.pack("<Q", len(data))
        self._f.write(header + struct.pack("<I", _masked_crc(header)) + data + struct.pack("<I", _masked_crc(data)))

    def add_scalar(self, tag, value, step):
        self._record(_event(time.time(), step=step, tag=tag, value=value))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def _read_varint(buf, pos):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, pos


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wire == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError("wire type %d" % wire)
        yield field, wire, v


def read_scalars(path):
    """[(tag, value, step, wall_time)] of an event file; raises ValueError on a checksum mismatch."""
    data = open(path, "rb").read()
    pos, out = 0, []
    while pos < len(data):
        header = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", header)
        if struct.unpack("<I", data[pos + 8:pos + 12])[0] != _masked_crc(header):
            raise ValueError("length checksum mismatch at byte %d" % pos)
        rec = data[pos + 12:pos + 12 + n]
        if struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] != _masked_crc(rec):
            raise ValueError("record checksum mismatch at byte %d" % pos)
        pos += 16 + n
        wall, step, summary = 0.0, 0, None
        for field, _wire, v in _fields(rec):
            if field == 1:
                (wall,) = struct.unpack("<d", v)
            elif field == 2:
                step = v - (1 << 64) if v >> 63 else v
            elif field == 5:
                summary = v
        if summary is None:
            continue
        for field, _wire, val in _fields(summary):
            if field != 1:
                continue
            tag, simple = None, None
            for f2, _w2, v2 in _fields(val):
                if f2 == 1:
                    tag = v2.decode()
                elif f2 == 2:
                    (simple,) = struct.unpack("<f", v2)
            if tag is not None and simple is not None:
                out.append((tag, simple, step, wall))
    return out
