// ref_linearize_api.cpp -- TEST INFRASTRUCTURE (never linked into the product).  The linear side of the object BA, edge by edge, as the reference's
// vendored g2o has it, cut out of the reference at build time (oracle/ref_shim/extract_ref.py -> oracle/_ref/extracted_lin_*.inc):
//   * BaseBinaryEdge::linearizeOplus / linearizeOplusXi / linearizeOplusXj / constructQuadraticForm (Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-320:
//     central differences with delta 1e-9 through push / oplus / computeError / pop, and the quadratic form with and without a robust kernel, in both
//     block layouts), BaseUnaryEdge::linearizeOplus / constructQuadraticForm (base_unary_edge.hpp:43-123), BaseEdge::chi2 / robustInformation
//     (base_edge.h:58-61, 96-102);
//   * the vertex and edge classes of Optimizer::BundleAdjustment / LocalBACameraPointObjects WHOLE: VertexSBAPointXYZ (types_sba.h:40-57), VertexSE3Expmap,
//     EdgeSE3ProjectXYZ, EdgeStereoSE3ProjectXYZ and the two pose-only edges of PoseOptimization (types_six_dof_expmap.h:59-99, 164-290) with their
//     out-of-line linearizeOplus / cam_project (types_six_dof_expmap.cpp), VertexCuboidFixScale, EdgeSE3CuboidFixScaleProj,
//     EdgePointCuboidOnlyObjectFixScale (include/g2o_Object.h:257-283, 325-338, 511-535; src/g2o_Object.cpp:87-128, 336-354).
// They compile against stand-ins for BaseVertex / BaseEdge / BaseBinaryEdge / BaseUnaryEdge that hold the members those texts name (estimate and its
// backup stack, the vertex's own A and b instead of maps into the solver's blocks, the edge's Hessian block in either layout) and against
// oracle/ref_shim/eigen_mini for Eigen's fixed-size interface.  Huber's weights come from the reference's RobustKernelHuber::robustify through
// ref_huber_robustify (ref_levenberg_api.cpp).  tests/test_ref_pins.py holds the oracle's build_system (ba_oracle.cpp) against ref_ba_linearize.
#include <algorithm>
#include <stack>
#include <vector>

#include "../oracle.h"
#include "ref_g2o_types.hpp"

extern "C" void ref_huber_robustify(double e, double delta, double *rho3);

namespace g2o {
using namespace Eigen;
using namespace std;

#define OptimizableGraph OptimizableGraphLin // (ref_levenberg_api.cpp has its own stand-in of that name in this library)
struct JacobianWorkspace {};
class RobustKernel { // core/robust_kernel.h: what an edge asks of its kernel
  public:
    double delta = 1;
    void robustify(double e, Eigen::Vector3d &rho) const { double r[3]; ref_huber_robustify(e, delta, r); rho[0] = r[0]; rho[1] = r[1]; rho[2] = r[2]; }
};
struct VertexBase { // what base_multi_edge.hpp asks of an OptimizableGraph::Vertex
    virtual ~VertexBase() {}
    virtual int dimension() const = 0;
    virtual bool fixed() const = 0;
    virtual void push() = 0;
    virtual void pop() = 0;
    virtual void oplus(const double *v) = 0;
};
struct OptimizableGraph { typedef VertexBase Vertex; };

template <int D, typename T> class BaseVertex : public VertexBase { // core/base_vertex.h, core/optimizable_graph.h (Vertex): the members the cut-out texts use
  public:
    static const int Dimension = D;
    BaseVertex() {}
    const T &estimate() const { return _estimate; }
    void setEstimate(const T &et) { _estimate = et; }           // base_vertex.h:101
    int dimension() const override { return D; }
    void push() override { _backup.push(_estimate); }                    // base_vertex.h:89
    void pop() override { _estimate = _backup.top(); _backup.pop(); }    // base_vertex.h:90
    void oplus(const double *v) override { oplusImpl(v); }               // optimizable_graph.h: Vertex::oplus = oplusImpl + updateCache
    virtual void oplusImpl(const double *v) = 0;
    virtual void setToOriginImpl() = 0;
    bool fixed() const override { return _fixed; }
    void setFixed(bool f) { _fixed = f; }
    Matrix<double, D, 1> &b() { return _b; }                     // (the real b() / A() are maps into the solver's vector and diagonal block)
    Matrix<double, D, D> &A() { return _A; }
  protected:
    T _estimate;
    std::stack<T> _backup;
    bool _fixed = false;
    Matrix<double, D, 1> _b;
    Matrix<double, D, D> _A;
};

template <int D, typename E> class BaseEdge { // core/base_edge.h
  public:
    static const int Dimension = D;
    typedef E Measurement;
    typedef Matrix<double, D, 1> ErrorVector;
    typedef Matrix<double, D, D> InformationType;
    virtual ~BaseEdge() {}
#include "extracted_lin_edge_members.inc"
    const InformationType &information() const { return _information; }
    void setInformation(const InformationType &i) { _information = i; }
    void setMeasurement(const Measurement &m) { _measurement = m; }
    const ErrorVector &error() const { return _error; }
    RobustKernel *robustKernel() const { return _robustKernel; }
    void setRobustKernel(RobustKernel *k) { _robustKernel = k; }
    void setVertex(size_t i, VertexBase *v) { _vertices[i] = v; }
    virtual void computeError() = 0;
  protected:
    Measurement _measurement;
    InformationType _information;
    ErrorVector _error;
    RobustKernel *_robustKernel = nullptr;
    std::vector<VertexBase *> _vertices;
};

template <int D, typename E, typename VertexXi, typename VertexXj> class BaseBinaryEdge : public BaseEdge<D, E> { // core/base_binary_edge.h
  public:
    typedef VertexXi VertexXiType;
    typedef VertexXj VertexXjType;
    static const int Di = VertexXiType::Dimension;
    static const int Dj = VertexXjType::Dimension;
    typedef typename BaseEdge<D, E>::ErrorVector ErrorVector;
    typedef typename BaseEdge<D, E>::InformationType InformationType;
    typedef Matrix<double, D, Di> JacobianXiOplusType;
    typedef Matrix<double, D, Dj> JacobianXjOplusType;
    BaseBinaryEdge() { _vertices.resize(2); }
    using BaseEdge<D, E>::computeError;
    const JacobianXiOplusType &jacobianOplusXi() const { return _jacobianOplusXi; }
    const JacobianXjOplusType &jacobianOplusXj() const { return _jacobianOplusXj; }
    virtual void linearizeOplus();
    virtual void linearizeOplusXi();
    virtual void linearizeOplusXj();
    void constructQuadraticForm();
    void mapHessianMemory(bool rowMajor) { _hessianRowMajor = rowMajor; } // base_binary_edge.hpp:324-334 (the block itself is the edge's own here)
    bool _hessianRowMajor = false;
    Matrix<double, Di, Dj> _hessian;
    Matrix<double, Dj, Di> _hessianTransposed;
  protected:
    using BaseEdge<D, E>::_measurement;
    using BaseEdge<D, E>::_information;
    using BaseEdge<D, E>::_error;
    using BaseEdge<D, E>::_vertices;
    JacobianXiOplusType _jacobianOplusXi;
    JacobianXjOplusType _jacobianOplusXj;
};

template <int D, typename E, typename VertexXi> class BaseUnaryEdge : public BaseEdge<D, E> { // core/base_unary_edge.h
  public:
    typedef VertexXi VertexXiType;
    typedef typename BaseEdge<D, E>::ErrorVector ErrorVector;
    typedef typename BaseEdge<D, E>::InformationType InformationType;
    typedef Matrix<double, D, VertexXiType::Dimension> JacobianXiOplusType;
    BaseUnaryEdge() { _vertices.resize(1); }
    using BaseEdge<D, E>::computeError;
    const JacobianXiOplusType &jacobianOplusXi() const { return _jacobianOplusXi; }
    virtual void linearizeOplus();
    void constructQuadraticForm();
  protected:
    using BaseEdge<D, E>::_measurement;
    using BaseEdge<D, E>::_information;
    using BaseEdge<D, E>::_error;
    using BaseEdge<D, E>::_vertices;
    JacobianXiOplusType _jacobianOplusXi;
};

template <int D, typename E> class BaseMultiEdge : public BaseEdge<D, E> { // core/base_multi_edge.h (this fork adds analytical_jaco_id / linearizeOplusXid)
  public:
    typedef typename BaseEdge<D, E>::ErrorVector ErrorVector;
    typedef typename BaseEdge<D, E>::InformationType InformationType;
    typedef DynMat<double> JacobianType; // (the real one maps D x dim of a JacobianWorkspace)
    BaseMultiEdge() { analytical_jaco_id = -1; _dimension = D; }
    void resize(size_t n) { _vertices.resize(n); _jacobianOplus.resize(n); }
    void sizeJacobians() { for (size_t i = 0; i < _vertices.size(); i++) _jacobianOplus[i].resize(D, _vertices[i]->dimension()); } // linearizeOplus(JacobianWorkspace&), base_multi_edge.hpp:51-60
    using BaseEdge<D, E>::computeError;
    virtual void linearizeOplus();
    virtual void linearizeOplusXid(int variable_id);
    int analytical_jaco_id;
    std::vector<JacobianType> _jacobianOplus;
  protected:
    using BaseEdge<D, E>::_measurement;
    using BaseEdge<D, E>::_information;
    using BaseEdge<D, E>::_error;
    using BaseEdge<D, E>::_vertices;
    int _dimension;
};

#include "extracted_lin_core.inc"
#include "extracted_lin_multi.inc"
#include "extracted_lin_types.inc"
#include "extracted_lin_cpp.inc"
#include "extracted_lin_dyn.inc"

// (declared virtual by the classes above; the graph-file readers are not on the path)
bool VertexSBAPointXYZ::read(std::istream &) { return false; }
bool VertexSBAPointXYZ::write(std::ostream &) const { return false; }
bool VertexSE3Expmap::read(std::istream &) { return false; }
bool VertexSE3Expmap::write(std::ostream &) const { return false; }
} // namespace g2o

namespace {
g2o::SE3Quat se3(const double *v) { g2o::Vector7d x; for (int i = 0; i < 7; i++) x[i] = v[i]; return g2o::SE3Quat(x); }
template <int R, int C> void put(const Eigen::Matrix<double, R, C> &m, double *o) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o[i * C + j] = m(i, j); }
} // namespace

extern "C" {
// One pass of BlockSolver::buildSystem (block_solver.hpp:488-560: per active edge linearizeOplus + constructQuadraticForm, edges in insertion order = point
// observations, camera-cuboid, point-cuboid) over the oracle's problem at its initial estimates, after computeActiveErrors.  Which layout an edge's block
// has is the block solver's choice (block_solver.hpp:221-250): an edge whose first vertex is marginalised writes the transposed block (pose x landmark),
// an edge between two pose-like vertices writes (i, j) directly when hessianIndex(i) < hessianIndex(j) -- cameras come before cuboids.
// Outputs (row-major, any may be NULL): Hpp_diag P x 36 (non-fixed cameras, then cuboids), Hll L x 9, Hpl n_obs x 18 (6 x 3), Hcc n_cobs x 36 (camera x cuboid),
// b 6P + 3L, err / chi2 per edge (3 per observation with a zero third row for monocular ones, 4 per camera-cuboid edge, 3 per point-cuboid edge).
int ref_ba_linearize(const orc_ba_problem *p, double *Hpp_diag, double *Hll, double *Hpl, double *Hcc, double *b, double *err, double *chi2) {
    using namespace g2o;
    std::vector<VertexSE3Expmap *> cams(p->n_cams);
    std::vector<VertexSBAPointXYZ *> pts(p->n_points);
    std::vector<VertexCuboidFixScale *> cubs(p->n_cuboids);
    for (int i = 0; i < p->n_cams; i++) { cams[i] = new VertexSE3Expmap(); cams[i]->setEstimate(se3(p->cam_pose + (size_t)i * 7)); cams[i]->setFixed(p->cam_fixed[i] != 0); }
    for (int i = 0; i < p->n_points; i++) { pts[i] = new VertexSBAPointXYZ(); pts[i]->setEstimate(Eigen::Vector3d(p->points[i * 3], p->points[i * 3 + 1], p->points[i * 3 + 2])); }
    for (int i = 0; i < p->n_cuboids; i++) {
        VertexCuboidFixScale *v = cubs[i] = new VertexCuboidFixScale();
        cuboid c; c.pose = se3(p->cuboid_pose + (size_t)i * 7);
        for (int k = 0; k < 3; k++) c.scale[k] = p->cuboid_scale[i * 3 + k];
        v->setEstimate(c);
        const int fl = p->cuboid_flags[i];
        v->whether_fixrollpitch = fl & 1; v->whether_fixrotation = (fl & 2) != 0; v->whether_fixheight = (fl & 4) != 0;
        if (fl & 8) for (int k = 0; k < 3; k++) v->fixedscale[k] = p->cuboid_scale[i * 3 + k];
    }
    RobustKernel k_mono, k_stereo, k_obj;
    k_mono.delta = p->huber_mono; k_stereo.delta = p->huber_stereo; k_obj.delta = p->huber_obj;
    size_t eo = 0, co = 0;
    for (int o = 0; o < p->n_obs; o++) {
        const bool stereo = p->obs_ur && p->obs_ur[o] >= 0;
        VertexSE3Expmap *vc = cams[p->obs_cam[o]];
        double *hpl = Hpl ? Hpl + (size_t)o * 18 : nullptr;
        if (!stereo) {
            EdgeSE3ProjectXYZ e;
            e.setVertex(0, pts[p->obs_point[o]]); e.setVertex(1, vc);
            e.setMeasurement(Eigen::Vector2d(p->obs_uv[o * 2], p->obs_uv[o * 2 + 1]));
            e.setInformation(Eigen::Matrix<double, 2, 2>::Identity() * p->obs_inv_sigma2[o]); // Optimizer.cc: Eigen::Matrix2d::Identity() * invSigma2
            if (p->huber_mono > 0) e.setRobustKernel(&k_mono);
            e.fx = p->fx; e.fy = p->fy; e.cx = p->cx; e.cy = p->cy;
            e.mapHessianMemory(true);
            e.computeError();
            if (err) { err[eo] = e.error()[0]; err[eo + 1] = e.error()[1]; err[eo + 2] = 0; }
            if (chi2) chi2[co] = e.chi2();
            e.linearizeOplus(); e.constructQuadraticForm();
            if (hpl && !vc->fixed()) put(e._hessianTransposed, hpl);
        } else {
            EdgeStereoSE3ProjectXYZ e;
            e.setVertex(0, pts[p->obs_point[o]]); e.setVertex(1, vc);
            e.setMeasurement(Eigen::Vector3d(p->obs_uv[o * 2], p->obs_uv[o * 2 + 1], p->obs_ur[o]));
            e.setInformation(Eigen::Matrix<double, 3, 3>::Identity() * p->obs_inv_sigma2[o]);
            if (p->huber_stereo > 0) e.setRobustKernel(&k_stereo);
            e.fx = p->fx; e.fy = p->fy; e.cx = p->cx; e.cy = p->cy; e.bf = p->bf;
            e.mapHessianMemory(true);
            e.computeError();
            if (err) for (int k = 0; k < 3; k++) err[eo + k] = e.error()[k];
            if (chi2) chi2[co] = e.chi2();
            e.linearizeOplus(); e.constructQuadraticForm();
            if (hpl && !vc->fixed()) put(e._hessianTransposed, hpl);
        }
        eo += 3; co++;
    }
    for (int o = 0; o < p->n_cobs; o++) {
        EdgeSE3CuboidFixScaleProj e;
        VertexSE3Expmap *vc = cams[p->cobs_cam[o]];
        e.setVertex(0, vc); e.setVertex(1, cubs[p->cobs_cuboid[o]]);
        e.setMeasurement(Eigen::Vector4d(p->cobs_bbox[o * 4], p->cobs_bbox[o * 4 + 1], p->cobs_bbox[o * 4 + 2], p->cobs_bbox[o * 4 + 3]));
        Eigen::Matrix<double, 4, 4> info;
        for (int k = 0; k < 4; k++) info(k, k) = p->cobs_info[o * 4 + k]; // Optimizer.cc: inv_sigma.cwiseProduct(inv_sigma).asDiagonal()
        e.setInformation(info);
        if (p->huber_obj > 0) e.setRobustKernel(&k_obj);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) e.Kalib(i, j) = p->K[i * 3 + j];
        e.mapHessianMemory(false);
        e.computeError();
        if (err) for (int k = 0; k < 4; k++) err[eo + k] = e.error()[k];
        if (chi2) chi2[co] = e.chi2();
        e.linearizeOplus(); e.constructQuadraticForm();
        if (Hcc && !vc->fixed()) put(e._hessian, Hcc + (size_t)o * 36);
        eo += 4; co++;
    }
    for (int o = 0; o < p->n_pc; o++) {
        EdgePointCuboidOnlyObjectFixScale e;
        e.setVertex(0, cubs[p->pc_cuboid[o]]);
        for (int i = p->pc_offsets[o]; i < p->pc_offsets[o + 1]; i++) e.object_points.push_back(Eigen::Vector3d(p->pc_points[i * 3], p->pc_points[i * 3 + 1], p->pc_points[i * 3 + 2]));
        e.max_outside_margin_ratio = p->max_outside_margin_ratio;
        e.setInformation(Eigen::Matrix<double, 3, 3>::Identity());
        e.computeError();
        if (err) for (int k = 0; k < 3; k++) err[eo + k] = e.error()[k];
        if (chi2) chi2[co] = e.chi2();
        e.linearizeOplus(); e.constructQuadraticForm();
        eo += 3; co++;
    }
    int P = 0;
    for (int i = 0; i < p->n_cams; i++) if (!cams[i]->fixed()) { if (Hpp_diag) put(cams[i]->A(), Hpp_diag + (size_t)P * 36); if (b) put(cams[i]->b(), b + (size_t)P * 6); P++; }
    for (int i = 0; i < p->n_cuboids; i++) { if (Hpp_diag) put(cubs[i]->A(), Hpp_diag + (size_t)P * 36); if (b) put(cubs[i]->b(), b + (size_t)P * 6); P++; }
    for (int i = 0; i < p->n_points; i++) { if (Hll) put(pts[i]->A(), Hll + (size_t)i * 9); if (b) put(pts[i]->b(), b + (size_t)P * 6 + (size_t)i * 3); }
    for (auto *v : cams) delete v;
    for (auto *v : pts) delete v;
    for (auto *v : cubs) delete v;
    return P;
}

// One linearisation of Optimizer::PoseOptimization's graph (Optimizer.cc:253-472): a single VertexSE3Expmap and n pose-only edges (monocular, or stereo where
// ur[i] >= 0), information = inv_sigma2 * I, Huber with delta_mono / delta_stereo where > 0.  H 6 x 6, b 6, chi2 n.
void ref_pose_linearize(int n, const double *Xw, const double *uv, const double *ur, const double *inv_sigma2, double fx, double fy, double cx, double cy, double bf, const double *pose7,
                        double delta_mono, double delta_stereo, double *H, double *b, double *chi2) {
    using namespace g2o;
    VertexSE3Expmap v; v.setEstimate(se3(pose7));
    RobustKernel k_mono, k_stereo; k_mono.delta = delta_mono; k_stereo.delta = delta_stereo;
    for (int i = 0; i < n; i++) {
        if (!(ur && ur[i] >= 0)) {
            EdgeSE3ProjectXYZOnlyPose e;
            e.setVertex(0, &v);
            e.setMeasurement(Eigen::Vector2d(uv[i * 2], uv[i * 2 + 1]));
            e.setInformation(Eigen::Matrix<double, 2, 2>::Identity() * inv_sigma2[i]);
            if (delta_mono > 0) e.setRobustKernel(&k_mono);
            e.fx = fx; e.fy = fy; e.cx = cx; e.cy = cy;
            for (int k = 0; k < 3; k++) e.Xw[k] = Xw[i * 3 + k];
            e.computeError(); if (chi2) chi2[i] = e.chi2();
            e.linearizeOplus(); e.constructQuadraticForm();
        } else {
            EdgeStereoSE3ProjectXYZOnlyPose e;
            e.setVertex(0, &v);
            e.setMeasurement(Eigen::Vector3d(uv[i * 2], uv[i * 2 + 1], ur[i]));
            e.setInformation(Eigen::Matrix<double, 3, 3>::Identity() * inv_sigma2[i]);
            if (delta_stereo > 0) e.setRobustKernel(&k_stereo);
            e.fx = fx; e.fy = fy; e.cx = cx; e.cy = cy; e.bf = bf;
            for (int k = 0; k < 3; k++) e.Xw[k] = Xw[i * 3 + k];
            e.computeError(); if (chi2) chi2[i] = e.chi2();
            e.linearizeOplus(); e.constructQuadraticForm();
        }
    }
    put(v.A(), H); put(v.b(), b);
}

// The edges of Optimizer::LocalBACameraPointObjectsDynamic's graph (Optimizer.cc:1537-2573) at the problem's estimates: computeError of every edge type, and the
// Jacobians of the two three-vertex types -- EdgeDynamicPointCuboidCamera's own linearizeOplus and BaseMultiEdge::linearizeOplus (central differences) over
// EdgeObjectMotion.  Layouts as orc_badyn_errors / orc_badyn_edge_jacobians.
void ref_badyn_edges(const orc_badyn_problem *p, double *e_obs, double *e_dobs, double *e_mot, double *e_cobs, double *e_pc, double *e_ulp, double *J_dobs, double *J_mot) {
    using namespace g2o;
    std::vector<VertexSE3Expmap> cams(p->n_cams);
    std::vector<VertexCuboidFixScale> objs(p->n_objs);
    std::vector<VelocityPlanarVelocity> vels(p->n_vels);
    std::vector<VertexSBAPointXYZ> pts(p->n_points), dpts(p->n_dpoints);
    for (int i = 0; i < p->n_cams; i++) { cams[i].setEstimate(se3(p->cam_pose + (size_t)i * 7)); cams[i].setFixed(p->cam_fixed[i] != 0); }
    for (int i = 0; i < p->n_objs; i++) {
        cuboid c; c.pose = se3(p->obj_pose + (size_t)i * 7);
        for (int k = 0; k < 3; k++) c.scale[k] = p->obj_scale[i * 3 + k];
        objs[i].setEstimate(c);
        const int fl = p->obj_flags[i];
        objs[i].whether_fixrollpitch = fl & 1; objs[i].whether_fixrotation = (fl & 2) != 0; objs[i].whether_fixheight = (fl & 4) != 0;
        if (fl & 8) for (int k = 0; k < 3; k++) objs[i].fixedscale[k] = p->obj_scale[i * 3 + k];
    }
    for (int i = 0; i < p->n_vels; i++) vels[i].setEstimate(Eigen::Vector2d(p->vel[i * 2], p->vel[i * 2 + 1]));
    for (int i = 0; i < p->n_points; i++) { pts[i].setEstimate(Eigen::Vector3d(p->points[i * 3], p->points[i * 3 + 1], p->points[i * 3 + 2])); pts[i].setFixed(p->fix_points != 0); }
    for (int i = 0; i < p->n_dpoints; i++) { dpts[i].setEstimate(Eigen::Vector3d(p->dpoints[i * 3], p->dpoints[i * 3 + 1], p->dpoints[i * 3 + 2])); dpts[i].setFixed(p->fix_points != 0); }
    Eigen::Matrix3d K;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) K(i, j) = p->K[i * 3 + j];
    for (int o = 0; o < p->n_obs; o++) {
        double *e = e_obs + (size_t)o * 3;
        if (!(p->obs_ur && p->obs_ur[o] >= 0)) {
            EdgeSE3ProjectXYZ ed; ed.setVertex(0, &pts[p->obs_point[o]]); ed.setVertex(1, &cams[p->obs_cam[o]]);
            ed.setMeasurement(Eigen::Vector2d(p->obs_uv[o * 2], p->obs_uv[o * 2 + 1])); ed.fx = p->fx; ed.fy = p->fy; ed.cx = p->cx; ed.cy = p->cy;
            ed.computeError(); e[0] = ed.error()[0]; e[1] = ed.error()[1]; e[2] = 0;
        } else {
            EdgeStereoSE3ProjectXYZ ed; ed.setVertex(0, &pts[p->obs_point[o]]); ed.setVertex(1, &cams[p->obs_cam[o]]);
            ed.setMeasurement(Eigen::Vector3d(p->obs_uv[o * 2], p->obs_uv[o * 2 + 1], p->obs_ur[o])); ed.fx = p->fx; ed.fy = p->fy; ed.cx = p->cx; ed.cy = p->cy; ed.bf = p->bf;
            ed.computeError(); for (int k = 0; k < 3; k++) e[k] = ed.error()[k];
        }
    }
    for (int o = 0; o < p->n_dobs; o++) {
        EdgeDynamicPointCuboidCamera ed;
        ed.setVertex(0, &cams[p->dobs_cam[o]]); ed.setVertex(1, &objs[p->dobs_obj[o]]); ed.setVertex(2, &dpts[p->dobs_point[o]]);
        ed.setMeasurement(Eigen::Vector2d(p->dobs_uv[o * 2], p->dobs_uv[o * 2 + 1])); ed.Kalib = K;
        ed.sizeJacobians();
        ed.computeError(); e_dobs[o * 2] = ed.error()[0]; e_dobs[o * 2 + 1] = ed.error()[1];
        ed.linearizeOplus();
        for (int v = 0; v < 3; v++) for (int k = 0; k < 2; k++) for (int a = 0; a < 6; a++) J_dobs[(size_t)o * 36 + v * 12 + k * 6 + a] = a < ed._jacobianOplus[v].cols() ? ed._jacobianOplus[v](k, a) : 0.0;
    }
    for (int o = 0; o < p->n_mot; o++) {
        EdgeObjectMotion ed;
        ed.setVertex(0, &objs[p->mot_from[o]]); ed.setVertex(1, &objs[p->mot_to[o]]); ed.setVertex(2, &vels[p->mot_vel[o]]);
        ed.delta_t = p->mot_dt[o];
        ed.sizeJacobians();
        ed.computeError(); for (int k = 0; k < 3; k++) e_mot[o * 3 + k] = ed.error()[k];
        ed.linearizeOplus();
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) J_mot[(size_t)o * 54 + v * 18 + k * 6 + a] = a < ed._jacobianOplus[v].cols() ? ed._jacobianOplus[v](k, a) : 0.0;
    }
    for (int o = 0; o < p->n_cobs; o++) {
        EdgeSE3CuboidFixScaleProj ed; ed.setVertex(0, &cams[p->cobs_cam[o]]); ed.setVertex(1, &objs[p->cobs_obj[o]]);
        ed.setMeasurement(Eigen::Vector4d(p->cobs_bbox[o * 4], p->cobs_bbox[o * 4 + 1], p->cobs_bbox[o * 4 + 2], p->cobs_bbox[o * 4 + 3])); ed.Kalib = K;
        ed.computeError(); for (int k = 0; k < 4; k++) e_cobs[o * 4 + k] = ed.error()[k];
    }
    for (int o = 0; o < p->n_pc; o++) {
        EdgePointCuboidOnlyObjectFixScale ed; ed.setVertex(0, &objs[p->pc_obj[o]]);
        for (int i = p->pc_offsets[o]; i < p->pc_offsets[o + 1]; i++) ed.object_points.push_back(Eigen::Vector3d(p->pc_points[i * 3], p->pc_points[i * 3 + 1], p->pc_points[i * 3 + 2]));
        ed.max_outside_margin_ratio = p->pc_ratio;
        ed.computeError(); for (int k = 0; k < 3; k++) e_pc[o * 3 + k] = ed.error()[k];
    }
    for (int i = 0; i < p->n_dpoints; i++) {
        UnaryLocalPoint ed; ed.setVertex(0, &dpts[i]);
        for (int k = 0; k < 3; k++) ed.objectscale[k] = p->ulp_scale[k];
        ed.max_outside_margin_ratio = p->ulp_ratio;
        ed.computeError(); for (int k = 0; k < 3; k++) e_ulp[i * 3 + k] = ed.error()[k];
    }
}
}
