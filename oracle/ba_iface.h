/* oracle/ba_iface.h -- TEST INFRASTRUCTURE.  What the LM-schedule pin (oracle/ref_shim/ref_levenberg_api.cpp: the reference's own
 * OptimizationAlgorithmLevenberg::solve / SparseOptimizer::optimize over the oracle's pieces) asks of a BA oracle; ba_oracle.cpp serves it from its own BA,
 * badyn_oracle.cpp provides this interface for the dynamic-object BA (orc_badyn_open). */
#pragma once
struct orc_badyn_problem;
struct OrcBAIface {
    virtual ~OrcBAIface() {}
    virtual void compute_errors() = 0;
    virtual double robust_chi2() = 0;
    virtual void build_system() = 0;
    virtual int n_blocks() = 0;                       /* vertices with a Hessian block: pose-like ones first, then the marginalised ones */
    virtual int n_pose_blocks() = 0;
    virtual int block_dim(int block) = 0;
    virtual double hessian_diag(int block, int j) = 0;
    virtual bool solve(double lambda) = 0;
    virtual void update() = 0;
    virtual void push() = 0;
    virtual void pop() = 0;
    virtual void discard_top() = 0;
    virtual const double *x(long *n) = 0;             /* pose part, then landmark part */
    virtual const double *b() = 0;
    virtual void read(double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints) = 0;
};
OrcBAIface *orc_badyn_make_iface(const struct orc_badyn_problem *p); /* badyn_oracle.cpp */
