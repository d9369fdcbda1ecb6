"""Test infrastructure: CPU restatement of the reference render path (see tensorf_oracle.py)."""
