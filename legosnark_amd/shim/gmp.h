// Stub: /root/reference/src/utils/matrix.h:11 includes <gmp.h> but uses nothing from it
// on the hot path; the MI355X library needs no GMP.
#pragma once
