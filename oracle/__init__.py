"""ORACLE -- test infrastructure only (see oracle/model.py header).  Never imported by the product."""
