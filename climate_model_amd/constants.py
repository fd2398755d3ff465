"""Physical constants of the dynamical core.

Values are those of the reference (io_constants.py:15-20); they must stay
bit-identical because they enter every stencil.
"""
con_g = 9.81
con_rE = 6371000.0
con_omega = 7.292115E-5
con_Rd = 287.058
con_cp = 1005.0
con_kappa = con_Rd / con_cp
