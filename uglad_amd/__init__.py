"""uglad_amd: MI355X-native unrolled-GLAD hot path behind uGLAD's Python surface."""
