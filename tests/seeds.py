"""The seeds the differential tests draw their inputs from.  GROOVE_TEST_SEEDS sets how many each test plays (its own default otherwise) and
GROOVE_TEST_SEED_BASE where they start (0): a campaign is `GROOVE_TEST_SEEDS=300 GROOVE_TEST_SEED_BASE=1000 pytest -m gpu -k random`."""
import os


def drawn_seeds(default):
    base = int(os.environ.get("GROOVE_TEST_SEED_BASE", "0"))
    return range(base, base + int(os.environ.get("GROOVE_TEST_SEEDS", str(default))))
