"""Achieved parity maxima of the GPU tests, collected during a session and written by conftest.py to
``gpurun_out/parity_r06.json`` (the file the explicit per-configuration bounds of the tests are justified by)."""
RECORDS = []


def add(rec: dict) -> None:
    RECORDS.append(rec)
