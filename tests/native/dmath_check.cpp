// CPU check of csrc/dmath.h against libm (built and run by tests/test_dmath.py): prints, per function, the largest relative
// error in double and the number of inputs whose result ROUNDED TO FLOAT differs from libm's rounded to float.
#include "../../fastdeepqlearning_amd/csrc/dmath.h"
#include <cstdio>
#include <cstdlib>
#include <random>

template <typename F, typename G>
static void sweep(const char *name, F f, G ref, double lo, double hi, long n, bool logspace) {
  double worst = 0, at = 0;
  long fdiff = 0;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (long i = 0; i < n; ++i) {
    const double t = (i + (i & 1 ? U(rng) : 0.5)) / (double)n;
    const double x = logspace ? lo * std::pow(hi / lo, t) : lo + (hi - lo) * t;
    const double a = f(x), b = ref(x);
    const double err = b != 0 ? std::fabs(a - b) / std::fabs(b) : std::fabs(a);
    if (err > worst) { worst = err; at = x; }
    if ((float)a != (float)b) ++fdiff;
  }
  std::printf("%s max_rel_err %.3e at %.17g float_mismatches %ld of %ld\n", name, worst, at, fdiff, n);
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 4000000;
  using namespace fdql;
  sweep("exp", [](double x) { return dm_exp(x); }, [](double x) { return std::exp(x); }, -21.0, 3.0, n, false);
  sweep("exp_wide", [](double x) { return dm_exp(x); }, [](double x) { return std::exp(x); }, -80.0, 80.0, n, false);
  sweep("log", [](double x) { return dm_log(x); }, [](double x) { return std::log(x); }, 1e-9, 50.0, n, true);
  sweep("log_near_1", [](double x) { return dm_log(x); }, [](double x) { return std::log(x); }, 0.5, 2.0, n, false);
  sweep("log_1m", [](double x) { return dm_log(1.0 - x); }, [](double x) { return std::log(1.0 - x); }, 1e-12, 0.4, n, true);
  sweep("log_1p", [](double x) { return dm_log(1.0 + x); }, [](double x) { return std::log(1.0 + x); }, 1e-12, 0.4, n, true);
  sweep("tanh", [](double x) { return dm_tanh(x); }, [](double x) { return std::tanh(x); }, -12.0, 12.0, n, false);
  sweep("tanh_small", [](double x) { return dm_tanh(x); }, [](double x) { return std::tanh(x); }, 1e-12, 0.5, n, true);
  sweep("tanh_neg_small", [](double x) { return dm_tanh(-x); }, [](double x) { return std::tanh(-x); }, 1e-12, 0.5, n, true);
  sweep("tanh_big", [](double x) { return dm_tanh(x); }, [](double x) { return std::tanh(x); }, 8.0, 60.0, n / 4, false);
  // non-finite and out-of-range inputs follow libm: NaN in -> NaN out, exp(-inf) = 0, exp(+inf) = inf, tanh(+-inf) = +-1
  const double inf = INFINITY, nan = NAN;
  int bad = 0;
  bad += !(dm_exp(nan) != dm_exp(nan));
  bad += !(dm_tanh(nan) != dm_tanh(nan));
  bad += !(dm_exp(-inf) == 0.0) + !(dm_exp(inf) == inf) + !(dm_exp(-1e4) == 0.0) + !(dm_exp(1e4) == inf);
  bad += !(dm_tanh(inf) == 1.0) + !(dm_tanh(-inf) == -1.0) + !(dm_tanh(1e300) == 1.0);
  bad += !((float)dm_exp(-744.0) == (float)std::exp(-744.0)) + !(dm_exp(709.0) == std::exp(709.0) || std::fabs(dm_exp(709.0) / std::exp(709.0) - 1.0) < 5e-15);
  std::printf("nonfinite_failures %d\n", bad);
  return 0;
}
