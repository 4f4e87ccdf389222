"""Weights of the golden fixtures (tests/golden/): regenerated from oracle.encoder_ref.det_state_dict -- a counter-based
generator written out in the oracle, independent of any torch / NumPy random stream -- and checked against the sha256 the
generating script recorded.  A mismatch is a failure, never a skip: these fixtures are the only evidence pinned to the
reference's own classes."""
from oracle import encoder_ref


def golden_weights(meta, **kw):
    assert meta.get("gen") == "det", "fixture predates the deterministic weight generator: run tests/golden/make_golden.py"
    sd = encoder_ref.det_state_dict(seed=meta["seed"], n_layers=meta["n_layers"], ln_jitter=meta["ln_jitter"], **kw)
    got = encoder_ref.state_dict_sha256(sd)
    assert got == meta["checksum"], "deterministic weights differ from the ones the golden vectors were made with: %s" % got
    return sd
