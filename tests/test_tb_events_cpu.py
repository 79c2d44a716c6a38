"""synthsr_amd/tb_events.py: the TensorBoard event file training() writes in place of KC.TensorBoard (SynthSR/training.py:431).
Checked without TensorFlow: CRC-32C against the RFC 3720 vectors, the Event / Summary encoding against google.protobuf with the
message types declared here from TensorFlow's published field numbers, a write -> read round trip, and corruption detection."""
import os
import struct

import numpy as np
import pytest

from synthsr_amd import tb_events as tbe


def test_crc32c_known_answers():
    assert tbe.crc32c(b'123456789') == 0xE3069283
    assert tbe.crc32c(bytes(32)) == 0x8A9136AA                     # RFC 3720 B.4
    assert tbe.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tbe.crc32c(bytes(range(32))) == 0x46DD794E
    c = tbe.crc32c(b'abc')
    assert tbe.masked_crc32c(b'abc') == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _tf_event_class():
    """tensorflow.Event / Summary / Summary.Value with the field numbers of tensorflow/core/util/event.proto and
    framework/summary.proto, built at run time (protobuf is installed, TensorFlow is not)"""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='synthsr_test_event.proto', package='synthsr_test', syntax='proto3')
    val = fd.message_type.add(name='Value')
    val.field.add(name='tag', number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    val.field.add(name='simple_value', number=2, type=F.TYPE_FLOAT, label=F.LABEL_OPTIONAL)
    summ = fd.message_type.add(name='Summary')
    summ.field.add(name='value', number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name='.synthsr_test.Value')
    ev = fd.message_type.add(name='Event')
    ev.field.add(name='wall_time', number=1, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    ev.field.add(name='step', number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ev.field.add(name='file_version', number=3, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    ev.field.add(name='summary', number=5, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name='.synthsr_test.Summary')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName('synthsr_test.Event')
    if hasattr(message_factory, 'GetMessageClass'):
        return message_factory.GetMessageClass(desc)
    return message_factory.MessageFactory(pool).GetPrototype(desc)


def test_event_encoding_parses_with_protobuf():
    Event = _tf_event_class()
    e = Event()
    e.ParseFromString(tbe.encode_event(1234.5, file_version='brain.Event:2'))
    assert e.wall_time == 1234.5 and e.step == 0 and e.file_version == 'brain.Event:2'
    e = Event()
    e.ParseFromString(tbe.encode_event(99.25, step=300, scalars=[('loss', 0.125), ('val', -3.5e-7)]))
    assert e.wall_time == 99.25 and e.step == 300 and len(e.summary.value) == 2
    assert e.summary.value[0].tag == 'loss' and e.summary.value[0].simple_value == 0.125
    assert e.summary.value[1].tag == 'val' and e.summary.value[1].simple_value == np.float32(-3.5e-7)
    # and the other way round: what protobuf serialises is what the reader understands
    e2 = Event(wall_time=5.0, step=7)
    v = e2.summary.value.add()
    v.tag, v.simple_value = 'loss', 2.5
    blob = e2.SerializeToString()
    assert blob == tbe.encode_event(5.0, step=7, scalars=[('loss', 2.5)])


def test_event_file_round_trip_and_corruption(tmp_path):
    with tbe.EventFileWriter(str(tmp_path), now=1700000000.5, hostname='box') as w:
        path = w.path
        losses = [0.5, 0.25, float(np.float32(1 / 3))]
        for epoch, l in enumerate(losses):
            w.add_scalar('loss', l, epoch, wall_time=1700000001.0 + epoch)
    assert os.path.basename(path) == 'events.out.tfevents.1700000000.box'
    ev = tbe.read_events(path)
    assert ev[0]['file_version'] == 'brain.Event:2' and ev[0]['wall_time'] == 1700000000.5
    assert [e['step'] for e in ev[1:]] == [0, 1, 2]
    assert [e['scalars'] for e in ev[1:]] == [[('loss', l)] for l in losses]
    # record framing: length | masked crc of the length | data | masked crc of the data
    raw = open(path, 'rb').read()
    n, = struct.unpack('<Q', raw[:8])
    assert struct.unpack('<I', raw[8:12])[0] == tbe.masked_crc32c(raw[:8])
    assert struct.unpack('<I', raw[12 + n:16 + n])[0] == tbe.masked_crc32c(raw[12:12 + n])
    # appending to an existing log (a resumed run in the same second) does not repeat the version record
    with tbe.EventFileWriter(str(tmp_path), now=1700000000.9, hostname='box') as w:
        w.add_scalar('loss', 0.1, 3)
    ev = tbe.read_events(path)
    assert len(ev) == 5 and sum('file_version' in e for e in ev) == 1 and ev[-1]['step'] == 3
    bad = bytearray(raw)
    bad[20] ^= 1
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(ValueError):
        tbe.read_events(path)
    open(path, 'wb').write(raw[:-3])
    with pytest.raises(ValueError):
        tbe.read_events(path)
