"""pdmp3_read's read-ahead is invisible: random call sequences on a parse-only handle (no GPU: return codes and byte
counts) against the oracle's restatement of the reference's streaming API."""
import pytest

import stream_replay
from oracle.oracle import OracleStream


@pytest.mark.parametrize("seed", range(12))
def test_call_sequences_match_the_reference_api(oracle, seed):
    from pdmp3_amd import api
    for name, mp3 in stream_replay.streams():
        dec = api.Decoder(parse_only=True)
        orc = OracleStream(oracle)
        try:
            stream_replay.replay(1000 * seed + len(name), mp3, dec, orc, compare_pcm=False)
        finally:
            dec.close()
            orc.close()
