"""CPU fp64 oracle (test infrastructure). Import only from tests/, bench.py cpu_baseline, smoke()."""
