// libwf_lde.so, unit 6 of 6 -- the DEEP composition polynomial on resident commitments (deep.hpp holds kernels and host code).
#include "wf_internal.hpp"

#include "kernels.hpp"
#include "fri_kernels.hpp"
#include "seg_kernels.hpp"

#include "deep.hpp"
