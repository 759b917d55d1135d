"""Measurement and maintenance tools (none of this is the product: legion_amd/ is); bench.py imports tools.bench_legs."""
