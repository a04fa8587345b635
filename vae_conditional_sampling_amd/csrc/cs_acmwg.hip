// AC-MWG kernels (history ring, fused accept/reject + impute) -- see cs_acmwg entry points.
#include "cs_mlp.h"
