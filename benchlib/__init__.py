"""benchlib -- the parts of bench.py (the repository's benchmark contract lives in bench.py's docstring)."""
