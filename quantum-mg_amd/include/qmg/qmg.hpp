// qmg.hpp -- umbrella header of the device facade.
#ifndef QMG_HPP
#define QMG_HPP
#include "qmg_device.hpp"
#include "lattice2d.hpp"
#include "cshift2d.hpp"
#include "stencil2d.hpp"
#include "operators.hpp"
#include "transfer.hpp"
#include "coarse.hpp"
#include "krylov.hpp"
#include "multigrid.hpp"
#include "batch.hpp"
#include "u1.hpp"
#include "reductions.hpp"
#endif
