"""TensorBoard event files without TensorFlow: what `KC.TensorBoard(log_dir=<model_dir>/logs, histogram_freq=0, ...)` leaves
behind for SynthSR's training loop (SynthSR/training.py:424-431) -- one scalar per epoch and metric: tag = the Keras log name
('loss'), step = the 0-based epoch index (Keras 2.3.1 callbacks/tensorboard_v1.py: `_write_logs(logs, epoch)`), in a file
`events.out.tfevents.<unix time>.<hostname>`.  (The graph dump of `write_graph=True` describes a TensorFlow graph that does not
exist here and is not written.)

File format (public, stable since TF 0.x): a sequence of TFRecords
    uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)       (little endian)
with masked_crc(x) = rotr15(crc32c(x)) + 0xa282ead8, each `data` a serialised `tensorflow.Event` protobuf:
    Event   { double wall_time = 1; int64 step = 2; oneof { string file_version = 3; Summary summary = 5; } }
    Summary { repeated Value value = 1; }      Value { string tag = 1; float simple_value = 2; }
The first record carries file_version = "brain.Event:2".  Only these fields are written; `read_events` parses them back (and
verifies both checksums of every record) for the tests and for resuming a log."""
import os
import socket
import struct
import time

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1      # Castagnoli polynomial, reflected
            tab.append(c)
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data):
    tab, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(v):
    v &= (1 << 64) - 1          # int64 two's complement, as protobuf encodes negative values
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(number, payload):
    return _varint((number << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time, step=0, file_version=None, scalars=None):
    """serialised tensorflow.Event: scalars = [(tag, value), ...] -> one Summary with one Value each"""
    ev = _varint((1 << 3) | 1) + struct.pack('<d', float(wall_time))
    if step:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if scalars is not None:
        summary = b''
        for tag, value in scalars:
            val = _field_bytes(1, str(tag).encode()) + _varint((2 << 3) | 5) + struct.pack('<f', float(value))
            summary += _field_bytes(1, val)
        ev += _field_bytes(5, summary)
    return ev


def _record(data):
    head = struct.pack('<Q', len(data))
    return head + struct.pack('<I', masked_crc32c(head)) + data + struct.pack('<I', masked_crc32c(data))


class EventFileWriter:
    """with EventFileWriter(log_dir) as w: w.add_scalar('loss', value, step)"""

    def __init__(self, log_dir, now=None, hostname=None):
        os.makedirs(log_dir, exist_ok=True)
        now = time.time() if now is None else now
        host = socket.gethostname() if hostname is None else hostname
        self.path = os.path.join(log_dir, 'events.out.tfevents.%010d.%s' % (int(now), host))
        self._f = open(self.path, 'ab')
        if self._f.tell() == 0:
            self._f.write(_record(encode_event(now, file_version='brain.Event:2')))
            self._f.flush()

    def add_scalars(self, scalars, step, wall_time=None):
        self._f.write(_record(encode_event(time.time() if wall_time is None else wall_time, step=step, scalars=list(scalars))))
        self._f.flush()

    def add_scalar(self, tag, value, step, wall_time=None):
        self.add_scalars([(tag, value)], step, wall_time)

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _read_varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _parse(buf):
    """{field number: [raw values]} of one protobuf message (varint -> int, 64-bit / 32-bit -> bytes, length-delimited -> bytes)"""
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            val, pos = bytes(buf[pos:pos + n]), pos + n
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.setdefault(num, []).append(val)
    return out


def read_events(path):
    """[{'wall_time', 'step', 'file_version' | 'scalars': [(tag, value)]}] of an event file; raises ValueError on a bad checksum
    or a truncated record"""
    data = open(path, 'rb').read()
    events, pos = [], 0
    while pos < len(data):
        if pos + 12 > len(data):
            raise ValueError('truncated record header at byte %d' % pos)
        head = data[pos:pos + 8]
        n, = struct.unpack('<Q', head)
        if struct.unpack('<I', data[pos + 8:pos + 12])[0] != masked_crc32c(head):
            raise ValueError('bad length checksum at byte %d' % pos)
        body = data[pos + 12:pos + 12 + n]
        if len(body) != n or pos + 16 + n > len(data):
            raise ValueError('truncated record at byte %d' % pos)
        if struct.unpack('<I', data[pos + 12 + n:pos + 16 + n])[0] != masked_crc32c(body):
            raise ValueError('bad data checksum at byte %d' % pos)
        pos += 16 + n
        msg = _parse(body)
        step = msg.get(2, [0])[0]
        ev = {'wall_time': struct.unpack('<d', msg[1][0])[0] if 1 in msg else 0.0,
              'step': step - (1 << 64) if step >= (1 << 63) else step}
        if 3 in msg:
            ev['file_version'] = msg[3][0].decode()
        if 5 in msg:
            ev['scalars'] = []
            for raw in _parse(msg[5][0]).get(1, []):
                v = _parse(raw)
                ev['scalars'].append((v[1][0].decode(), struct.unpack('<f', v[2][0])[0] if 2 in v else 0.0))
        events.append(ev)
    return events
