"""ORACLE package: test infrastructure only (see stcn_oracle.py header)."""
