"""snappier_amd/multidevice.py on the one-GPU box: devices = [0, 0] -- two contexts, two worker threads, two streams on one GPU.  What the
ranges produce, concatenated on the host, must be the single-device bytes, i.e. the oracle's (SnappyStreamCompressor.cs:166-230)."""
import numpy as np
import pytest

import oracle as O
import snappier_amd as S
from conftest import read_testdata

pytestmark = pytest.mark.gpu


def _data():
    html = read_testdata("html")
    rng = np.random.default_rng(11)
    noise = bytes(rng.integers(0, 256, 3 * 65536, dtype=np.uint8))            # chunks stored raw (type 0x01)
    return html * 40 + noise + html[:12345]                                    # ~4.3 MiB: 66 chunks, the last one partial


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_ranges_over_several_contexts_give_the_single_device_bytes(devices):
    data = _data()
    with S.MultiDeviceCodec(devices, min_chunks_per_range=4) as md:
        framed = md.frame_encode(data)
        assert len(md.last_directory) == len(devices) and sum(md.last_directory) == len(framed)
        assert framed == O.frame_encode(data)
        assert md.frame_decode(framed) == data
        assert len(md.last_directory) == len(devices)
        block = md.compress(data)
        assert block == O.compress(data)
        assert S.Snappy.DecompressToArray(block) == data


def test_the_first_bad_chunk_in_stream_order_decides():
    data = _data()
    with S.MultiDeviceCodec([0, 0], min_chunks_per_range=4) as md:
        framed = bytearray(md.frame_encode(data))
        # corrupt a payload byte in the SECOND range's first chunk and one in the FIRST range's last chunk: the first range's error is reported
        from snappier_amd.multidevice import chunk_table
        table, _end = chunk_table(np.frombuffer(bytes(framed), dtype=np.uint8))
        k = len(table) // 2
        framed[table[k + 2][0] + 20] ^= 0xFF
        framed[table[3][0] + 20] ^= 0xFF
        with pytest.raises(S.InvalidDataException) as e_multi:
            md.frame_decode(bytes(framed))
        with pytest.raises(S.InvalidDataException) as e_single:
            S.frame_decode(bytes(framed))
        assert str(e_multi.value) == str(e_single.value)
