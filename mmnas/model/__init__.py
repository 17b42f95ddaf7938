"""`mmnas.model`: every module of the reference's `mmnas/model/` has its MI355X counterpart here; nothing is looked up
in the integrator's checkout."""
