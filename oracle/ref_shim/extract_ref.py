#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes oracle/_ref/extracted.inc: the text of selected function definitions of the reference, cut out of the
reference's files where they lie under /root/reference at BUILD time (oracle/_ref/ is git-ignored: no reference source enters the
repository).  The functions are found by their signature and cut by brace matching; oracle/ref_shim/ref_extract_api.cpp includes the result.
Used for files whose other functions need Eigen / g2o / the SLAM classes, which are absent here."""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_ref")

WANT = [  # (file, signatures, output include)
    ("detect_3d_cuboid/src/object_3d_util.cpp", ["void atan2_vector(", "void merge_break_lines(", "double box_edge_sum_dists(", "double box_edge_alignment_angle_error(", "void fuse_normalize_scores_v2("], "extracted_cuboid.inc"),
    ("detect_3d_cuboid/src/matrix_utils.cpp", ["void fast_RemoveRow(", "void sort_indexes(const Eigen::VectorXd &vec, std::vector<int> &idx, int top_k)", "T normalize_to_pi(T angle)"], "extracted_cuboid.inc"),
    ("orb_object_slam/src/ORBmatcher.cc", ["void ORBmatcher::ComputeThreeMaxima(", "int ORBmatcher::DescriptorDistance("], "extracted_orb.inc"),
    # members of class g2o::cuboid (pasted into a class of the same name with the same two data members, ref_g2o_api.cpp) ...
    ("orb_object_slam/include/g2o_Object.h", ["cuboid exp_update(const Vector9d &update)", "Vector9d cube_log_error(const cuboid &newone) const",
                                               "Vector9d min_log_error(const cuboid &newone, bool print_details = false) const", "cuboid rotate_cuboid(double yaw_angle) const",
                                               "cuboid transform_from(const SE3Quat &Twc) const", "cuboid transform_to(const SE3Quat &Twc) const",
                                               "Matrix4d similarityTransform() const", "Matrix3Xd compute3D_BoxCorner() const",
                                               "Vector4d projectOntoImageRect(const SE3Quat &campose_cw, const Matrix3d &Kalib) const",
                                               "Vector4d projectOntoImageBbox(const SE3Quat &campose_cw, const Matrix3d &Kalib) const"], "extracted_g2o_members.inc"),
    # the two coordinate helpers those call (detect_3d_cuboid/matrix_utils.h)
    ("detect_3d_cuboid/src/matrix_utils.cpp", ["Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> real_to_homo_coord(const Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> &pts_in)",
                                               "Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> homo_to_real_coord(const Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> &pts_homo_in)"], "extracted_g2o_utils.inc"),
    # ... and the free / out-of-line functions of g2o_Object.cpp that need nothing but SE3Quat and fixed-size vectors
    ("orb_object_slam/src/g2o_Object.cpp", ["SE3Quat exptwist_norollpitch(const Vector6d &update)", "Vector3d cuboid::point_boundary_error("], "extracted_g2o_cpp.inc"),
    # geometry of the cuboid proposals: vanishing points, their supporting edges, the boundary / intersection predicates of the corner construction,
    # 2D corners -> 3D cuboid (compiled against eigdyn.hpp, ref_geom_api.cpp)
    ("detect_3d_cuboid/src/matrix_utils.cpp", ["T normalize_to_pi(T angle)",
                                               "Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> real_to_homo_coord(const Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> &pts_in)",
                                               "Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> homo_to_real_coord(const Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> &pts_homo_in)",
                                               "Eigen::Matrix<T, Eigen::Dynamic, 1> homo_to_real_coord_vec(const Eigen::Matrix<T, Eigen::Dynamic, 1> &pts_homo_in)",
                                               "void quat_to_euler_zyx(const Eigen::Quaternion<T> &q, T &roll, T &pitch, T &yaw)",
                                               "Eigen::Matrix<T, 3, 3> euler_zyx_to_rot(const T &roll, const T &pitch, const T &yaw)",
                                               "void linespace(T starting, T ending, T step, std::vector<T> &res)",
                                               "void fast_RemoveRow(", "void sort_indexes(const Eigen::VectorXd &vec, std::vector<int> &idx, int top_k)"], "extracted_geom.inc"),
    ("detect_3d_cuboid/src/object_3d_util.cpp", ["Matrix4d similarityTransformation(const cuboid &cube_obj)", "Matrix3Xd compute3D_BoxCorner(const cuboid &cube_obj)",
                                                 "bool check_inside_box(const Vector2d &pt, const Vector2d &box_left_top, const Vector2d &box_right_bottom)",
                                                 "void align_left_right_edges(MatrixXd &all_lines)", "void atan2_vector(",
                                                 "void smooth_jump_angles(const VectorXd &raw_angles, VectorXd &new_angles)",
                                                 "Vector2d seg_hit_boundary(const Vector2d &pt_start, const Vector2d &pt_end, const Vector4d &line_segment2)",
                                                 "Vector2d lineSegmentIntersect(const Vector2d &pt1_start, const Vector2d &pt1_end, const Vector2d &pt2_start, const Vector2d &pt2_end,",
                                                 "void merge_break_lines(",
                                                 "Eigen::MatrixXd VP_support_edge_infos(Eigen::MatrixXd &VPs, Eigen::MatrixXd &edge_mid_pts, Eigen::VectorXd &edge_angles,",
                                                 "double box_edge_sum_dists(", "double box_edge_alignment_angle_error(", "void fuse_normalize_scores_v2(",
                                                 "void ray_plane_interact(const MatrixXd &rays, const Eigen::Vector4d &plane, MatrixXd &intersections)",
                                                 "void plane_hits_3d(const Matrix4d &transToWolrd, const Matrix3d &invK, const Vector4d &plane_sensor, MatrixXd pixels, Matrix3Xd &pts_3d_world)",
                                                 "Vector4d get_wall_plane_equation(const Vector3d &gnd_seg_pt1, const Vector3d &gnd_seg_pt2)",
                                                 "void getVanishingPoints(const Matrix3d &KinvR, double yaw_esti, Vector2d &vp_1, Vector2d &vp_2, Vector2d &vp_3)",
                                                 "void change_2d_corner_to_3d_object(const MatrixXd &box_corners_2d_float, const Vector3d &configs, const Vector4d &ground_plane_sensor,"], "extracted_geom.inc"),
    # ... and detect_cuboid itself, with the two setters it relies on
    ("detect_3d_cuboid/src/box_proposal_detail.cpp", ["void detect_3d_cuboid::set_calibration(const Matrix3d &Kalib)", "void detect_3d_cuboid::set_cam_pose(const Matrix4d &transToWolrd)",
                                                      "void detect_3d_cuboid::detect_cuboid(const cv::Mat &rgb_img, const Matrix4d &transToWolrd, const MatrixXd &obj_bbox_coors,"], "extracted_geom.inc"),
    # the ORB matcher's three window searches with what they call, and the Frame grid they search (compiled against stand-ins for Frame / MapPoint, ref_match_api.cpp)
    ("orb_object_slam/src/ORBmatcher.cc", ["const int ORBmatcher::TH_HIGH = 100;", "const int ORBmatcher::TH_LOW = 50;", "const int ORBmatcher::HISTO_LENGTH = 30;",
                                           "int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint *> &vpMapPoints, const float th)", "float ORBmatcher::RadiusByViewingCos(const float &viewCos)",
                                           "int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)",
                                           "int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)",
                                           "int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint *> &vpMapPointMatches)",
                                           "int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12)",
                                           "bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2)",
                                           "int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,",
                                           "int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th)",
                                           "void ORBmatcher::ComputeThreeMaxima(", "int ORBmatcher::DescriptorDistance("], "extracted_match.inc"),
    ("orb_object_slam/src/KeyFrame.cc", ["vector<size_t> KeyFrame::GetFeaturesInArea(const float &x, const float &y, const float &r) const",
                                         "bool KeyFrame::IsInImage(const float &x, const float &y) const"], "extracted_match.inc"),
    ("orb_object_slam/src/Frame.cc", ["void Frame::AssignFeaturesToGrid()", "vector<size_t> Frame::GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel, const int maxLevel) const",
                                      "bool Frame::PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY)"], "extracted_match.inc"),
    # g2o's Levenberg-Marquardt schedule and the optimiser's iteration loop (compiled against stand-ins for Solver / SparseOptimizer, ref_levenberg_api.cpp)
    ("orb_object_slam/Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp", ["OptimizationAlgorithmLevenberg::OptimizationAlgorithmLevenberg(Solver* solver) :",
                                                                                      "OptimizationAlgorithm::SolverResult OptimizationAlgorithmLevenberg::solve(int iteration, bool online)",
                                                                                      "double OptimizationAlgorithmLevenberg::computeLambdaInit() const",
                                                                                      "double OptimizationAlgorithmLevenberg::computeScale() const",
                                                                                      "void OptimizationAlgorithmLevenberg::printVerbose(std::ostream& os) const"], "extracted_levenberg.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/core/sparse_optimizer.cpp", ["int SparseOptimizer::optimize(int iterations, bool online)"], "extracted_levenberg.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp", ["void RobustKernelHuber::setDelta(double delta)", "void RobustKernelHuber::robustify(double e, Eigen::Vector3d& rho) const"], "extracted_huber.inc"),
    # the BA's linear side per edge: g2o's numeric Jacobians and quadratic forms (templates of base_binary_edge.hpp / base_unary_edge.hpp), the two one-liners of
    # base_edge.h they call, the vertex / edge classes of the object BA with their out-of-line members (compiled against stand-ins for BaseVertex / BaseEdge,
    # ref_linearize_api.cpp)
    ("orb_object_slam/Thirdparty/g2o/g2o/core/base_binary_edge.hpp", ["void BaseBinaryEdge<D, E, VertexXiType, VertexXjType>::constructQuadraticForm()",
                                                                      "void BaseBinaryEdge<D, E, VertexXiType, VertexXjType>::linearizeOplus()",
                                                                      "void BaseBinaryEdge<D, E, VertexXiType, VertexXjType>::linearizeOplusXi()",
                                                                      "void BaseBinaryEdge<D, E, VertexXiType, VertexXjType>::linearizeOplusXj()"], "extracted_lin_core.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/core/base_unary_edge.hpp", ["void BaseUnaryEdge<D, E, VertexXiType>::constructQuadraticForm()",
                                                                     "void BaseUnaryEdge<D, E, VertexXiType>::linearizeOplus()"], "extracted_lin_core.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/core/base_edge.h", ["virtual double chi2() const", "InformationType robustInformation(const Eigen::Vector3d& rho)"], "extracted_lin_edge_members.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/types/types_sba.h", ["class VertexSBAPointXYZ : public BaseVertex<3, Vector3d>"], "extracted_lin_types.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/types/types_six_dof_expmap.h", ["class  VertexSE3Expmap : public BaseVertex<6, SE3Quat>{",
                                                                         "class  EdgeSE3ProjectXYZ: public  BaseBinaryEdge<2, Vector2d, VertexSBAPointXYZ, VertexSE3Expmap>{",
                                                                         "class  EdgeStereoSE3ProjectXYZ: public  BaseBinaryEdge<3, Vector3d, VertexSBAPointXYZ, VertexSE3Expmap>{",
                                                                         "class  EdgeSE3ProjectXYZOnlyPose: public  BaseUnaryEdge<2, Vector2d, VertexSE3Expmap>{",
                                                                         "class  EdgeStereoSE3ProjectXYZOnlyPose: public  BaseUnaryEdge<3, Vector3d, VertexSE3Expmap>{"], "extracted_lin_types.inc"),
    ("orb_object_slam/include/g2o_Object.h", ["class VertexCuboidFixScale : public BaseVertex<6, cuboid>",
                                              "class EdgeSE3CuboidFixScaleProj : public BaseBinaryEdge<4, Vector4d, VertexSE3Expmap, VertexCuboidFixScale>",
                                              "class EdgePointCuboidOnlyObjectFixScale : public BaseUnaryEdge<3, Vector3d, VertexCuboidFixScale>"], "extracted_lin_types.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/types/types_sba.cpp", ["VertexSBAPointXYZ::VertexSBAPointXYZ() : BaseVertex<3, Vector3d>()"], "extracted_lin_cpp.inc"),
    ("orb_object_slam/Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp", ["Vector2d project2d(const Vector3d& v)  {", "VertexSE3Expmap::VertexSE3Expmap() : BaseVertex<6, SE3Quat>() {",
                                                                           "EdgeSE3ProjectXYZ::EdgeSE3ProjectXYZ() : BaseBinaryEdge<2, Vector2d, VertexSBAPointXYZ, VertexSE3Expmap>() {",
                                                                           "void EdgeSE3ProjectXYZ::linearizeOplus() {", "Vector2d EdgeSE3ProjectXYZ::cam_project(const Vector3d & trans_xyz) const{",
                                                                           "Vector3d EdgeStereoSE3ProjectXYZ::cam_project(const Vector3d & trans_xyz, const float &bf) const{",
                                                                           "EdgeStereoSE3ProjectXYZ::EdgeStereoSE3ProjectXYZ() : BaseBinaryEdge<3, Vector3d, VertexSBAPointXYZ, VertexSE3Expmap>() {",
                                                                           "void EdgeStereoSE3ProjectXYZ::linearizeOplus() {",
                                                                           "void EdgeSE3ProjectXYZOnlyPose::linearizeOplus() {", "Vector2d EdgeSE3ProjectXYZOnlyPose::cam_project(const Vector3d & trans_xyz) const{",
                                                                           "Vector3d EdgeStereoSE3ProjectXYZOnlyPose::cam_project(const Vector3d & trans_xyz) const{",
                                                                           "void EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus() {"], "extracted_lin_cpp.inc"),
    ("orb_object_slam/src/g2o_Object.cpp", ["void VertexCuboidFixScale::oplusImpl(const double *update_)", "void EdgeSE3CuboidFixScaleProj::computeError()",
                                            "void EdgePointCuboidOnlyObjectFixScale::computeError()"], "extracted_lin_cpp.inc"),
    # ... and the three-vertex edges of the dynamic-object BA with g2o's numeric Jacobians for them
    ("orb_object_slam/Thirdparty/g2o/g2o/core/base_multi_edge.hpp", ["void BaseMultiEdge<D, E>::linearizeOplus()", "void BaseMultiEdge<D, E>::linearizeOplusXid(int variable_id)"], "extracted_lin_multi.inc"),
    ("orb_object_slam/include/g2o_Object.h", ["class VelocityPlanarVelocity : public BaseVertex<2, Vector2d>", "class UnaryLocalPoint : public BaseUnaryEdge<3, Vector3d, VertexSBAPointXYZ>",
                                              "class EdgeDynamicPointCuboidCamera : public BaseMultiEdge<2, Vector2d>", "class EdgeObjectMotion : public BaseMultiEdge<3, Vector3d>"], "extracted_lin_dyn.inc"),
    ("orb_object_slam/src/g2o_Object.cpp", ["void EdgeDynamicPointCuboidCamera::computeError()", "void EdgeDynamicPointCuboidCamera::linearizeOplus()", "void EdgeObjectMotion::computeError()",
                                            "void UnaryLocalPoint::computeError()"], "extracted_lin_dyn.inc"),
    # the graph-level optimisation functions, whole, with the conversions they call (compiled against the reference's own g2o and stand-ins for the map classes,
    # ref_graph_api.cpp / slam_graph_standins.hpp)
    ("orb_object_slam/src/Converter.cc", ["g2o::SE3Quat Converter::toSE3Quat(const cv::Mat &cvT)", "cv::Mat Converter::toCvMat(const g2o::SE3Quat &SE3)",
                                          "cv::Mat Converter::toCvMat(const Eigen::Matrix<double, 4, 4> &m)", "cv::Mat Converter::toCvMat(const Eigen::Matrix<double, 3, 1> &m)",
                                          "Eigen::Matrix<double, 3, 1> Converter::toVector3d(const cv::Mat &cvVector)"], "extracted_graph_conv.inc"),
    ("orb_object_slam/src/Optimizer.cc", ["void Optimizer::BundleAdjustment(const vector<KeyFrame *> &vpKFs, const vector<MapPoint *> &vpMP,", "int Optimizer::PoseOptimization(Frame *pFrame)",
                                          "void Optimizer::LocalBACameraPointObjects(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera, bool fixPoint)",
                                          "void Optimizer::LocalBACameraPointObjectsDynamic(KeyFrame *pKF, bool *pbStopFlag, Map *pMap, bool fixCamera, bool fixPoint)",
                                          "void Optimizer::GlobalBundleAdjustemnt(Map *pMap, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust)",
                                          "void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap)"], "extracted_graph.inc"),
    # object association: Tracking::AssociateCuboids with the MapObject / MapPoint vote bookkeeping it drives
    ("orb_object_slam/src/MapObject.cc", ["long int MapObject::getIncrementedIndex()", "vector<MapPoint *> MapObject::GetUniqueMapPoints()", "int MapObject::NumUniqueMapPoints()",
                                          "void MapObject::AddUniqueMapPoint(MapPoint *pMP, int obs_num)", "void MapObject::EraseUniqueMapPoint(MapPoint *pMP, int obs_num)",
                                          "vector<MapPoint *> MapObject::GetPotentialMapPoints()", "void MapObject::AddPotentialMapPoint(MapPoint *pMP)",
                                          "bool MapObject::check_whether_valid_object(int own_point_thre)", "void MapObject::SetAsLandmark()",
                                          "void MapObject::MergeIntoLandmark(MapObject *otherLocalObject)", "void MapObject::addObservation(KeyFrame *pKF, size_t idx)"], "extracted_graph_map.inc"),
    ("orb_object_slam/src/MapPoint.cc", ["void MapPoint::AddObjectObservation(MapObject *obj)"], "extracted_graph_map.inc"),
    ("orb_object_slam/src/Tracking.cc", ["void Tracking::AssociateCuboids(KeyFrame *pKF)"], "extracted_graph.inc"),
    # the LBD descriptor: BinaryDescriptor's compute path (the rest of binary_descriptor.cpp is the EDLine detector, which CubeSLAM does not use)
    ("line_lbd/libs/binary_descriptor.cpp", ["static const int combinations[32][2] =", "BinaryDescriptor::Params::Params()", "BinaryDescriptor::BinaryDescriptor( const BinaryDescriptor::Params &parameters ) :",
                                             "BinaryDescriptor::~BinaryDescriptor()", "static inline int get2Pow( int i )", "void BinaryDescriptor::computeGaussianPyramid( const Mat& image, const int numOctaves )",
                                             "void BinaryDescriptor::computeSobel( const cv::Mat& image, const int numOctaves )", "unsigned char BinaryDescriptor::binaryConversion( float* f1, float* f2 )",
                                             "void BinaryDescriptor::compute( const Mat& image, CV_OUT CV_IN_OUT std::vector<KeyLine>& keylines, CV_OUT Mat& descriptors,",
                                             "void BinaryDescriptor::computeImpl( const Mat& imageSrc, std::vector<KeyLine>& keylines, Mat& descriptors, bool returnFloatDescr,",
                                             "int BinaryDescriptor::computeLBD( ScaleLines &keyLines, bool useDetectionData )"], "extracted_lbd.inc"),
]


def cut(text, sig):
    i = text.index(sig)
    if sig.endswith(";"):  # a one-line definition
        return sig
    start = text.rfind("\n", 0, i) + 1
    if text[max(0, start - 20):start].strip().startswith("template") or text[text.rfind("\n", 0, start - 1) + 1:start].startswith("template"):
        start = text.rfind("\n", 0, start - 1) + 1  # keep the `template <class T>` line
    j = text.index("{", i)
    depth, k = 0, j
    while True:
        if text[k] == "{":
            depth += 1
        elif text[k] == "}":
            depth -= 1
            if depth == 0:
                break
        k += 1
    return text[start:k + 1]


outs = {}
for rel, sigs, dst in WANT:
    text = open(os.path.join(REF, rel), encoding="utf-8", errors="replace").read()
    out = outs.setdefault(dst, ["// generated by oracle/ref_shim/extract_ref.py from %s -- not tracked" % REF])
    for s in sigs:
        out.append("// ---- %s : %s" % (rel, s))
        out.append(cut(text, s) + (";" if s.rstrip().endswith("=") or s.lstrip().startswith("class ") else ""))
os.makedirs(OUT, exist_ok=True)
for dst, out in outs.items():
    open(os.path.join(OUT, dst), "w").write("\n".join(out) + "\n")
