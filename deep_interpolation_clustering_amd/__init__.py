"""MI355X-native hot path of deep temporal interpolation + clustering.

Hand-written HIP (gfx950) kernels behind a C ABI (``include/dic_hip.h`` -> ``libdic_hip.so``) for
the RBF-kernel temporal interpolation, the RBF de-interpolation + masked reconstruction loss, the
DEC soft assignment / KL objective and k-means, exposed through the reference's own module surface
(``interpolation_layer``, ``rbf``, ``dec``, ``clustering_interp``, ``pretrain_interp``, the two
trainers and the p1-p4 drivers).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = '0.1.0'
