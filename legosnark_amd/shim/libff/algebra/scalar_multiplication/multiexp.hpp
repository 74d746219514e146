// libff-compatible include path (see ../../lsa_libff.hpp): LegoSNARK's sources include
// <libff/algebra/scalar_multiplication/multiexp.hpp>; everything is provided by the single shim header.
#pragma once
#include "../../lsa_libff.hpp"
