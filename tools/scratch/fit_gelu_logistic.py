"""fit of the shared-exponential GELU form of csrc/common.h (gelu_and_code2): Phi(x) ~ 1 / (1 + exp(-x (a + b x^2 [+ c x^4]))), minimax over
|x| <= 8 of max(|GELU error|, |GELU' error| / 2) against the erf form (video_swin.py:66 nn.GELU, HF hidden_act="gelu")."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf
x = np.linspace(-8, 8, 40001)
Phi = 0.5 * (1 + erf(x / np.sqrt(2))); phi = np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
gelu, dgelu = x * Phi, Phi + x * phi
def model(p, x):
    a, b, c = p
    s = 1 / (1 + np.exp(-x * (a + b * x * x + c * x ** 4)))
    return x * s, s + x * s * (1 - s) * (a + 3 * b * x * x + 5 * c * x ** 4)
def cost(p):
    g, dg = model(p, x)
    return max(np.abs(g - gelu).max(), 0.5 * np.abs(dg - dgelu).max())
r2 = minimize(lambda p: cost([p[0], p[1], 0.0]), [1.5957, 0.0713], method="Nelder-Mead", options=dict(xatol=1e-9, fatol=1e-12))
g, dg = model([r2.x[0], r2.x[1], 0.0], x)
print("2-term (shipped)", r2.x, "max |GELU err|", np.abs(g - gelu).max(), "max |GELU' err|", np.abs(dg - dgelu).max())
r3 = minimize(cost, [1.5957, 0.0713, 0.0], method="Nelder-Mead", options=dict(xatol=1e-9, fatol=1e-12, maxiter=20000))
g, dg = model(r3.x, x)
print("3-term", r3.x, "max |GELU err|", np.abs(g - gelu).max(), "max |GELU' err|", np.abs(dg - dgelu).max())
