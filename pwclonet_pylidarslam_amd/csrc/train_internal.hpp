// Launchers shared between the training-path translation units (conv1x1.hip <-> batchnorm.hip); not part of the forward
// path's sources (build.source_stamp()).
#pragma once
#include "common.hpp"

namespace pwclo {
// dgamma[ch] = sum of the partials' second entries, dbeta[ch] = sum of the first (conv1x1.hip's input-gradient epilogue)
void bn_backward_finish_launch(int c, int nsplit, const double *partial, float *dgamma, float *dbeta);
}  // namespace pwclo
