// TEST INFRASTRUCTURE: stands where the reference's Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h is included from (this directory comes first on the include
// path of oracle/Makefile.ref's g2o build).  The reference's file is nothing but a wrapper of Eigen's sparse Cholesky (SimplicialLDLT with an AMD ordering computed
// once) -- third-party arithmetic that is not in this image -- so the class keeps its interface (linear_solver_eigen.h:50-131: init, solve, blockOrdering,
// setBlockOrdering, writeDebug) and solves the same symmetric system, given by its upper blocks, with a dense L D L^T.  An exact factorisation in another
// elimination order: the solution agrees with Eigen's to round-off, which is what the graph-level pins (tests/test_ref_graph_pins.py) are held to.
#ifndef G2O_LINEAR_SOLVER_EIGEN_H
#define G2O_LINEAR_SOLVER_EIGEN_H

#include <Eigen/Cholesky>
#include <Eigen/Core>
#include <vector>

#include "Thirdparty/g2o/g2o/core/batch_stats.h"
#include "Thirdparty/g2o/g2o/core/linear_solver.h"

namespace g2o {
template <typename MatrixType> class LinearSolverEigen : public LinearSolver<MatrixType> {
  public:
    LinearSolverEigen() : LinearSolver<MatrixType>(), _init(true), _blockOrdering(false), _writeDebug(false) {}
    virtual ~LinearSolverEigen() {}
    virtual bool init() { _init = true; return true; }
    bool solve(const SparseBlockMatrix<MatrixType> &A, double *x, double *b) {
        const int n = A.cols();
        Eigen::MatrixXd H = Eigen::MatrixXd::Zero(n, n);
        for (size_t c = 0; c < A.blockCols().size(); ++c) { // the upper triangle, like fillSparseMatrix (:199-228)
            const int cb = A.colBaseOfBlock(c);
            const typename SparseBlockMatrix<MatrixType>::IntBlockMap &column = A.blockCols()[c];
            for (typename SparseBlockMatrix<MatrixType>::IntBlockMap::const_iterator it = column.begin(); it != column.end(); ++it) {
                const int rb = A.rowBaseOfBlock(it->first);
                const MatrixType &m = *(it->second);
                for (int cc = 0; cc < m.cols(); ++cc)
                    for (int rr = 0; rr < m.rows(); ++rr) {
                        if (rb + rr > cb + cc) break;
                        H(rb + rr, cb + cc) = m(rr, cc); H(cb + cc, rb + rr) = m(rr, cc);
                    }
            }
        }
        _init = false;
        Eigen::LDLT<Eigen::MatrixXd> chol(H);
        if (chol.info() != Eigen::Success || !chol.isPositive()) return false; // SimplicialLDLT::factorize reports a non-positive pivot the same way (:103-110)
        Eigen::VectorXd::MapType xx(x, n);
        Eigen::VectorXd::ConstMapType bb(b, n);
        xx = chol.solve(bb);
        return true;
    }
    bool blockOrdering() const { return _blockOrdering; }
    void setBlockOrdering(bool blockOrdering) { _blockOrdering = blockOrdering; }
    virtual bool writeDebug() const { return _writeDebug; }
    virtual void setWriteDebug(bool b) { _writeDebug = b; }

  protected:
    bool _init, _blockOrdering, _writeDebug;
};
} // namespace g2o
#endif
