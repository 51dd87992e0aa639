"""ORACLE package - test infrastructure only; the product (tenscalc_amd/) never imports it."""
