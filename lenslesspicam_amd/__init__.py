"""
lenslesspicam_amd -- MI355X-native engine for LenslessPiCam's iterative deconvolution hot path
(ADMM / gradient descent / Nesterov / FISTA over FFT convolution with a fixed PSF).

The classes mirror ``lensless.recon``'s plugin API so existing scripts can switch imports:

    from lenslesspicam_amd import ADMM, FISTA, GradientDescent, NesterovGradientDescent, RealFFTConvolve2D
"""
from .admm import ADMM, apply_admm
from .gd import (FISTA, GradientDescent, GradientDescentUpdate, NesterovGradientDescent,
                 apply_gradient_descent, non_neg)
from .recon import ReconstructionAlgorithm
from .rfft_convolve import RealFFTConvolve2D
from .unrolled_admm import UnrolledADMM
from .unrolled_fista import UnrolledFISTA
from . import metric, prep  # noqa: F401  (on-device evaluation metrics; raw-frame preparation)

__all__ = ["ADMM", "FISTA", "GradientDescent", "GradientDescentUpdate", "NesterovGradientDescent",
           "RealFFTConvolve2D", "ReconstructionAlgorithm", "UnrolledADMM", "UnrolledFISTA", "apply_admm", "apply_gradient_descent", "non_neg"]
__version__ = "0.1.0"
