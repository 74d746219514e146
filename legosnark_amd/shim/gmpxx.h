// Stub: see gmp.h (matrix.h:12).
#pragma once
