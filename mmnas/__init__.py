"""Alias package: `mmnas.*` import paths of the reference resolve to mmnas_amd (the MI355X implementation)."""
