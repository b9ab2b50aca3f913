"""The host's chain-round policy (csrc/tc_chain.hpp, tc_encode_host.hpp) at a size where its own trigger fires (2^22 members
>= 2^20): a periodic text is done in three rounds with ONE chain round; a Fibonacci word -- repetitive, not periodic: the
chains are short -- makes the chain rounds back off (at most three attempts); a text with a long run of one symbol inside
random text takes a chain round on SPARSE ranks.  Each result is compared with the same encode without chain rounds
(TC_SA_CHAIN=0: the plain doubling, pinned to the oracle elsewhere) and decoded."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fib(n):
    a, b = np.array([65], np.uint8), np.array([65, 67], np.uint8)
    while len(b) < n:
        a, b = b, np.concatenate([b, a])
    return b[:n]


def _texts():
    n = 1 << 22
    rng = np.random.default_rng(7)
    periodic = np.resize(rng.integers(0, 4, 4096).astype(np.uint8) + 65, n)
    # random text with ONE run of 1.5 * 2^20 'N's: the run's members are the tied set -- over 2^20 of them (the trigger's
    # floor), under an eighth of the 2^24 suffixes (the ranks stay sparse)
    n2 = 1 << 24
    gaps = rng.integers(0, 4, n2).astype(np.uint8) + 65
    gaps[n2 // 3:n2 // 3 + 3 * (1 << 19)] = 78
    return {"periodic": periodic, "fibonacci": _fib(n), "run_in_random": gaps}


@pytest.mark.parametrize("name", ["periodic", "fibonacci", "run_in_random"])
def test_chain_round_policy(name):
    import textcomp
    t = _texts()[name].tobytes()
    with textcomp.Context(0) as ctx:
        os.environ.pop("TC_SA_CHAIN", None)
        blk = ctx.encode(t)
        st = ctx.stats()
        rounds, chain = int(st.rounds), int(st.chain_rounds)
        os.environ["TC_SA_CHAIN"] = "0"
        try:
            ref = ctx.encode(t)
            st0 = ctx.stats()
        finally:
            os.environ.pop("TC_SA_CHAIN", None)
        assert int(st0.chain_rounds) == 0
        assert blk["primary"] == ref["primary"] and blk["final_list"].tolist() == ref["final_list"].tolist()
        assert np.array_equal(blk["run_count"], ref["run_count"]) and np.array_equal(blk["run_value"], ref["run_value"])
        assert ctx.decode(blk) == t
        if name == "periodic":
            assert chain == 1 and rounds == 3 and int(st0.rounds) > 10, (rounds, chain, int(st0.rounds))
        elif name == "fibonacci":
            assert 1 <= chain <= 3 and rounds == int(st0.rounds), (rounds, chain, int(st0.rounds))
        else:
            assert chain >= 1 and rounds < int(st0.rounds), (rounds, chain, int(st0.rounds))
