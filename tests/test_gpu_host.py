"""The host-layer cases of test_host_layer.py again, this time on the product path: C++ host layer -> C ABI -> HIP kernels
on the MI355X.  (Imported test functions are collected here and pick up this module's `eng` fixture.)"""
import pytest

from plz4_amd import host
from test_host_layer import (  # noqa: F401
    test_example_new_writer, test_example_new_reader, test_the_works_written_byte_exact, test_writer_matrix,
    test_uncompressable_blocks_are_stored, test_empty_input_sync_vs_async, test_flush_makes_short_blocks,
    test_progress_and_read_offset, test_writer_sink_failures, test_short_read, test_content_crc,
    test_block_crc_and_size_overflow, test_concatenated_and_skippable_frames, test_read_small_chunks_matches_write_to,
    test_corrupt_block_payload_is_lz4_corrupted, test_block_api,
    test_writer_dict_and_linked_roundtrip, test_dictionary_makes_small_payloads_smaller)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = host.hip_engine(0)
    yield e
    e.close()


def test_every_level_and_mode_is_built(eng):
    """The counterpart of test_unsupported_modes_fail_loudly: the HIP engine has every compressor
    compress.NewCompressorFactory can hand out (compress/compress.go:32-80), so none of these is ErrUnsupported."""
    from plz4_amd import synth
    payload = synth.text(200000, seed=77).tobytes()
    d = synth.text(5000, seed=78).tobytes()
    for kw in (dict(level=9, block_linked=True), dict(level=12, block_linked=True), dict(level=2, dictionary=d),
               dict(level=5, block_linked=True, dictionary=d)):
        w = host.Writer(eng, parallel=1, block_size=host.BlockIdx64KB, **kw)
        assert w.write(payload)[1] == 0 and not w.close()
        n, out, err = host.Reader(eng, w.output(), dictionary=kw.get("dictionary")).write_to()
        assert not err and out == payload
