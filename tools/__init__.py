"""Build, profile and probe scripts; `bench_legs` holds the auxiliary legs of bench.py."""
