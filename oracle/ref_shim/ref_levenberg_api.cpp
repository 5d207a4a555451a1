// TEST INFRASTRUCTURE (never linked into the product): g2o's Levenberg-Marquardt as the reference's vendored copy has it --
// OptimizationAlgorithmLevenberg::OptimizationAlgorithmLevenberg / solve / computeLambdaInit / computeScale
// (orb_object_slam/Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:43-56, 61-164, 166-180, 182-189) and the iteration loop
// SparseOptimizer::optimize (sparse_optimizer.cpp:354-419) -- cut out of the reference at build time (oracle/_ref/extracted_levenberg.inc) and compiled
// against stand-ins for Solver, SparseOptimizer, OptimizableGraph::Vertex and the property map whose methods hand the work to the ORACLE's pieces
// (orc_ba_open ... orc_ba_read, ba_oracle.cpp): residuals, the quadratic form, the Schur solve and the state stack are the oracle's, the schedule --
// lambda's initial value, the trial loop, the gain ratio and its scale, how lambda grows and shrinks, the three ways to stop -- is the reference's.
// tests/test_ref_pins.py compares a run driven this way with orc_ba_optimize, whose loop restates that schedule.
#include <cassert>
#include <cmath>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include "../oracle.h"

#define FIXED(s) s
namespace g2o {
using namespace std;
inline double get_monotonic_time() { return 0; }
inline bool g2o_isfinite(double x) { return std::isfinite(x); }
struct G2OBatchStatistics {
    int iteration = 0, levenbergIterations = 0; size_t numEdges = 0, numVertices = 0;
    double chi2 = 0, timeResiduals = 0, timeQuadraticForm = 0, timeLinearSolution = 0, timeUpdate = 0, timeIteration = 0;
    static G2OBatchStatistics *globalStats() { return nullptr; }
    static void setGlobalStats(G2OBatchStatistics *) {}
};
template <typename T> struct Property { T v; const T &value() const { return v; } void setValue(const T &x) { v = x; } };
struct PropertyMap { template <typename P, typename T> P *makeProperty(const std::string &, const T &v) { P *p = new P(); p->setValue(v); return p; } };

class SparseOptimizer;
struct OptimizableGraph {
    struct Vertex {
        orc_ba_handle *h; int block, dim;
        int dimension() const { return dim; }
        double hessian(int i, int j) const { assert(i == j); return orc_ba_hessian_diag(h, block, i); }
    };
    typedef std::vector<Vertex *> VertexContainer;
};
class Solver { // core/solver.h: what the algorithm asks of the linear side
  public:
    orc_ba_handle *h = nullptr; SparseOptimizer *opt = nullptr;
    double lambda = 0; int solves = 0;
    bool buildStructure(bool = false) { return true; }
    bool buildSystem() { orc_ba_build_system(h); return true; }
    bool setLambda(double l, bool = false) { lambda = l; return true; }
    bool solve() { solves++; return orc_ba_solve(h, lambda) != 0; } // (the oracle adds lambda while it reduces: nothing to restore)
    void restoreDiagonal() {}
    const double *x() const { return orc_ba_x(h, nullptr); }
    const double *b() const { return orc_ba_b(h); }
    size_t vectorSize() const { long n = 0; orc_ba_x(h, &n); return (size_t)n; }
    SparseOptimizer *optimizer() const { return opt; }
    bool schur() { return true; }
};
class OptimizationAlgorithm {
  public:
    enum SolverResult { Terminate = 2, OK = 1, Fail = -1 }; // optimization_algorithm.h:49
    virtual ~OptimizationAlgorithm() {}
    virtual bool init(bool online = false) = 0;
    virtual SolverResult solve(int iteration, bool online = false) = 0;
    virtual void printVerbose(std::ostream &) const {}
    SparseOptimizer *_optimizer = nullptr;
    PropertyMap _properties;
};
class OptimizationAlgorithmWithHessian : public OptimizationAlgorithm {
  public:
    explicit OptimizationAlgorithmWithHessian(Solver *solver) : _solver(solver) {}
    virtual bool init(bool = false) { return true; }
    Solver *_solver;
};
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithmWithHessian { // optimization_algorithm_levenberg.h
  public:
    explicit OptimizationAlgorithmLevenberg(Solver *solver);
    virtual SolverResult solve(int iteration, bool online = false);
    virtual void printVerbose(std::ostream &os) const;
    double currentLambda() const { return _currentLambda; }
    int levenbergIteration() { return _levenbergIterations; }
  protected:
    Property<int> *_maxTrialsAfterFailure;
    Property<double> *_userLambdaInit;
    double _currentLambda, _tau, _goodStepLowerScale, _goodStepUpperScale, _ni;
    int _levenbergIterations, _nBad;
    double computeLambdaInit() const;
    double computeScale() const;
};
class SparseOptimizer { // core/sparse_optimizer.h: what optimize() and the algorithm touch
  public:
    orc_ba_handle *h = nullptr;
    OptimizableGraph::VertexContainer _ivMap;
    OptimizationAlgorithm *_algorithm = nullptr;
    std::vector<G2OBatchStatistics> _batchStatistics;
    bool _computeBatchStatistics = false;
    std::vector<int> _activeEdges, _activeVertices;
    bool terminate() { return false; }
    bool verbose() const { return false; }
    void preIteration(int) {}
    void postIteration(int) {}
    void computeActiveErrors() { orc_ba_compute_errors(h); }
    double activeRobustChi2() const { return orc_ba_robust_chi2(h); }
    void push() { orc_ba_push(h); }
    void pop() { orc_ba_pop(h); }
    void discardTop() { orc_ba_discard_top(h); }
    void update(const double *) { orc_ba_update(h); } // (the oracle applies its own x, the vector Solver::x() points at)
    const OptimizableGraph::VertexContainer &indexMapping() const { return _ivMap; }
    int optimize(int iterations, bool online = false);
};
#include "extracted_levenberg.inc"
} // namespace g2o

// RobustKernelHuber::setDelta / robustify (core/robust_kernel_impl.cpp:65-69, 78-91): rho, rho', rho'' of a squared error
namespace EigenK { struct Vector3d { double v[3]; double &operator[](int i) { return v[i]; } }; }
namespace g2o {
namespace Eigen = EigenK;
class RobustKernelHuber { // robust_kernel_impl.h (this copy of g2o keeps delta squared next to delta)
  public:
    void setDelta(double delta);
    void robustify(double e, Eigen::Vector3d &rho) const;
    double _delta = 1., dsqr = 1.;
};
#include "extracted_huber.inc"
} // namespace g2o

extern "C" {
// Optimizer::BundleAdjustment's `optimizer.optimize(nIterations)` (Optimizer.cc:232) over the oracle's problem.  Returns the iterations done; trials =
// linear solves, lambda_final as the algorithm leaves it.
int ref_ba_levenberg(const orc_ba_problem *p, int iterations, double *cam_pose_out, double *points_out, double *cuboid_pose_out, int *trials, double *lambda_final, double *chi2_final) {
    orc_ba_handle *h = orc_ba_open(p);
    g2o::SparseOptimizer opt; g2o::Solver solver;
    opt.h = h; solver.h = h; solver.opt = &opt;
    int P = 0, L = 0;
    orc_ba_sizes(h, &P, &L);
    std::vector<g2o::OptimizableGraph::Vertex> verts((size_t)P + L);
    for (int k = 0; k < P + L; k++) { verts[k] = g2o::OptimizableGraph::Vertex{h, k, k < P ? 6 : 3}; opt._ivMap.push_back(&verts[k]); }
    g2o::OptimizationAlgorithmLevenberg alg(&solver);
    alg._optimizer = &opt; opt._algorithm = &alg;
    const int done = opt.optimize(iterations);
    orc_ba_compute_errors(h);
    *chi2_final = orc_ba_robust_chi2(h);
    *trials = solver.solves; *lambda_final = alg.currentLambda();
    orc_ba_read(h, cam_pose_out, points_out, cuboid_pose_out);
    orc_ba_close(h);
    return done;
}
// The same schedule over the dynamic-object BA oracle's pieces (Optimizer::LocalBACameraPointObjectsDynamic's `optimizer.optimize(...)`, Optimizer.cc:2336, 2451):
// vertices are the non-fixed cameras, the objects (6), the velocities (2), then the marginalised points (3).
int ref_badyn_levenberg(const orc_badyn_problem *p, int iterations, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints, int *trials, double *lambda_final,
                        double *chi2_final) {
    orc_ba_handle *h = orc_badyn_open(p);
    g2o::SparseOptimizer opt; g2o::Solver solver;
    opt.h = h; solver.h = h; solver.opt = &opt;
    int P = 0, L = 0;
    orc_ba_sizes(h, &P, &L);
    std::vector<g2o::OptimizableGraph::Vertex> verts((size_t)P + L);
    for (int k = 0; k < P + L; k++) { verts[k] = g2o::OptimizableGraph::Vertex{h, k, orc_ba_block_dim(h, k)}; opt._ivMap.push_back(&verts[k]); }
    g2o::OptimizationAlgorithmLevenberg alg(&solver);
    alg._optimizer = &opt; opt._algorithm = &alg;
    const int done = opt.optimize(iterations);
    orc_ba_compute_errors(h);
    *chi2_final = orc_ba_robust_chi2(h);
    *trials = solver.solves; *lambda_final = alg.currentLambda();
    orc_badyn_read(h, cam_pose, obj_pose, vel, points, dpoints);
    orc_ba_close(h);
    return done;
}
void ref_huber_robustify(double e, double delta, double *rho3) {
    g2o::RobustKernelHuber k; k.setDelta(delta);
    EigenK::Vector3d r; k.robustify(e, r);
    rho3[0] = r[0]; rho3[1] = r[1]; rho3[2] = r[2];
}
}
