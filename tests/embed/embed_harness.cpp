// C++ embed harness: drives the Python mirror exactly the way DSP-SLAM's C++ does, without Eigen/OpenCV.
//
// Replays, from a NON-MAIN std::thread under PyGILState_Ensure (reference include/System.h:56-70), the call / cast
// sequence of   src/System.cc:90-99,152 (interpreter, sys.path, get_configs, get_decoder, GIL release),
//               src/LocalMapping.cc:38-40 (Optimizer / MeshExtractor construction),
//               src/LocalMapping_util.cc:109-110 (estimate_pose_cam_obj -> Matrix4f),
//               src/LocalMapping_util.cc:179-191 (reconstruct_object -> is_good / t_cam_obj / code),
//               src/LocalMapping_util.cc:391-413 (5-argument form with a warm-start code, loss, code_len),
//               src/LocalMapping_util.cc:194-196,426-428 (extract_mesh_from_code -> vertices MatrixXf / faces MatrixXi).
// With an argument "seq=<dir>": the call that follows get_decoder in the System constructor, src/System.cc:98 --
// `py::module::import("reconstruct").attr("get_sequence")(strSequencePath, pyCfg)` -- whose classes are the REFERENCE'S files, reached through
// the mirror package's __path__ (dsp_slam_amd/reconstruct/__init__.py); prints the class and the file it came from.
// With a 4th argument "mono": the MONOCULAR sequence of src/LocalMapping_util.cc:391-428 instead -- 5-argument call with the map object's
// 64-float vShapeCode, the 180-degree-yaw-flipped second call for an object that is not reconstructed yet, the `loss` comparison that
// picks one of the two, t_cam_obj -> Matrix4f, code_len read as int, the code cast to a FIXED-SIZE Vector<float,32> when code_len == 32
// (a size mismatch would throw cast_error there) and copied into the head of a zeroed 64-vector, and that 64-vector handed to
// extract_mesh_from_code.
// Eigen is column-major and pybind11's Eigen caster hands numpy Fortran-ordered float32 COPIES; the harness builds the
// same kind of arrays (py::array::f_style).  Inputs are read from an .npz the Python test wrote; results are printed as
// "key v0 v1 ..." lines for the test to compare.
#include <pybind11/embed.h>
#include <pybind11/numpy.h>

#include <cstdio>
#include <string>
#include <thread>
#include <vector>

namespace py = pybind11;
using farr = py::array_t<float, py::array::f_style | py::array::forcecast>;

static void print_arr(const char* key, const py::object& o) {
    py::array_t<float, py::array::c_style | py::array::forcecast> a(o);   // what .cast<Eigen::...>() would copy out
    std::printf("%s", key);
    const float* p = a.data();
    for (py::ssize_t i = 0; i < a.size(); ++i) std::printf(" %.9g", p[i]);
    std::printf("\n");
}

struct PyThreadStateLock {   // reference include/System.h:56-70
    PyThreadStateLock() { state = PyGILState_Ensure(); }
    ~PyThreadStateLock() { PyGILState_Release(state); }
    PyGILState_STATE state;
};

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: embed_harness <mirror_dir> <cfg.json> <inputs.npz> [mono] [seq=<sequence dir>]\n"); return 2; }
    const std::string mirror = argv[1], cfg_file = argv[2], npz = argv[3];
    bool mono = false;
    std::string seq_dir;
    for (int i = 4; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "mono") mono = true;
        else if (a.rfind("seq=", 0) == 0) seq_dir = a.substr(4);
    }
    std::setvbuf(stdout, nullptr, _IOLBF, 1 << 16);                            // whole lines, so that Python's own prints cannot land inside one
    py::initialize_interpreter();                                              // System.cc:90
    py::object pyCfg, pyDecoder;
    {
        py::module sys = py::module::import("sys");                           // System.cc:92
        sys.attr("path").attr("insert")(0, mirror);                            // the one change: the mirror goes first (or PYTHONPATH)
        sys.attr("path").attr("append")("./");                                 // System.cc:93
        py::module io_utils = py::module::import("reconstruct.utils");         // System.cc:94
        pyCfg = io_utils.attr("get_configs")(cfg_file);                        // System.cc:96
        pyDecoder = io_utils.attr("get_decoder")(pyCfg);                       // System.cc:97
        if (!seq_dir.empty()) {
            py::object pySequence = py::module::import("reconstruct").attr("get_sequence")(seq_dir, pyCfg);      // System.cc:98
            const std::string cls = py::str(py::type::of(pySequence).attr("__name__"));
            const std::string file = py::str(py::module::import("sys").attr("modules")[py::type::of(pySequence).attr("__module__")].attr("__file__"));
            std::printf("sequence %s %s\n", cls.c_str(), file.c_str());
        }
    }
    PyThreadState* main_state = PyEval_SaveThread();                           // System.cc:152: GIL released by the main thread

    int rc = 0;
    std::thread local_mapping([&] {                                            // LocalMapping runs in its own std::thread
        PyThreadStateLock lock;
        try {
            py::module optim = py::module::import("reconstruct.optimizer");   // LocalMapping.cc:38
            py::object pyOptimizer = optim.attr("Optimizer")(pyDecoder, pyCfg);
            py::object pyMeshExtractor = optim.attr("MeshExtractor")(pyDecoder, pyCfg.attr("optimizer").attr("code_len"), pyCfg.attr("voxels_dim"));
            pyOptimizer.attr("verbose") = false;
            py::object data = py::module::import("numpy").attr("load")(npz);
            auto F = [&](const char* k) { return farr(data[k]); };                // Fortran-ordered float32 copy, like the Eigen caster

            if (mono) {
                // ProcessDetectedObjects (monocular)                                      (LocalMapping_util.cc:391-428)
                py::module np = py::module::import("numpy");
                farr t_co = F("t_cam_obj");
                py::object shape_code = np.attr("zeros")(64, "float32");                 // pMO->vShapeCode: Vector<float,64>, zero before the first reconstruction
                py::object a = pyOptimizer.attr("reconstruct_object")(t_co, F("pts"), F("rays"), F("depth"), farr(shape_code));
                // flipped_Two.col(0) *= -1; flipped_Two.col(2) *= -1  ->  the same columns of SE3Tcw * Two
                farr t_flip(t_co.attr("copy")("F"));
                {
                    auto m = t_flip.mutable_unchecked<2>();
                    for (int r = 0; r < 4; ++r) { m(r, 0) = -m(r, 0); m(r, 2) = -m(r, 2); }
                }
                py::object b = pyOptimizer.attr("reconstruct_object")(t_flip, F("pts"), F("rays"), F("depth"), farr(shape_code));
                const float loss_a = a.attr("loss").cast<float>(), loss_b = b.attr("loss").cast<float>();     // read unconditionally, as the reference does (:405)
                std::printf("mono_loss %.9g %.9g\n", loss_a, loss_b);
                py::object pick = loss_a > loss_b ? b : a;
                std::printf("mono_pick %d\n", loss_a > loss_b ? 1 : 0);
                print_arr("mono_t_cam_obj", pick.attr("t_cam_obj"));                     // .cast<Eigen::Matrix4f>()
                const int code_len = pyOptimizer.attr("code_len").cast<int>();
                std::printf("code_len %d\n", code_len);
                py::array_t<float, py::array::c_style | py::array::forcecast> code_arr(pick.attr("code"));
                // Eigen::Vector<float, 32> / <float, 64> are fixed-size: pybind11's caster refuses any other length
                if (code_arr.ndim() != 1 || code_arr.shape(0) != (code_len == 32 ? 32 : 64)) throw std::runtime_error("code length does not fit the fixed-size Eigen vector");
                std::vector<float> code64(64, 0.f);
                for (py::ssize_t i = 0; i < code_arr.shape(0); ++i) code64[i] = code_arr.data()[i];
                print_arr("mono_code", pick.attr("code"));
                py::array_t<float> code64_arr(64, code64.data());
                py::object pyMesh = pyMeshExtractor.attr("extract_mesh_from_code")(code64_arr);     // the zero-padded 64-vector, also for a 32-D decoder
                py::array_t<float, py::array::c_style | py::array::forcecast> verts(pyMesh.attr("vertices"));
                py::array_t<int, py::array::c_style | py::array::forcecast> faces(pyMesh.attr("faces"));
                std::printf("mesh_shape %lld %lld %lld %lld\n", (long long)verts.shape(0), (long long)verts.shape(1), (long long)faces.shape(0), (long long)faces.shape(1));
                print_arr("mesh_vertices", pyMesh.attr("vertices"));
                return;
            }
            // GetNewObservations: pose-only, result cast to a 4x4 matrix           (LocalMapping_util.cc:109-110)
            py::object se3 = pyOptimizer.attr("estimate_pose_cam_obj")(F("pose_t_co_se3"), data["pose_scale"].cast<float>(), F("pose_pts"), F("pose_code"));
            print_arr("pose_only", se3);

            // CreateNewMapObjects: 4-argument form                                    (LocalMapping_util.cc:179-191)
            py::object obj = pyOptimizer.attr("reconstruct_object")(F("t_cam_obj"), F("pts"), F("rays"), F("depth"));
            const bool good = obj.attr("is_good").cast<bool>();
            std::printf("is_good %d\n", good ? 1 : 0);
            if (good) {
                print_arr("t_cam_obj", obj.attr("t_cam_obj"));
                print_arr("code", obj.attr("code"));
                // ProcessDetectedObjects: 5-argument warm start, loss as float, code_len as int (LocalMapping_util.cc:391-413)
                py::object obj2 = pyOptimizer.attr("reconstruct_object")(F("t_cam_obj"), F("pts"), F("rays"), F("depth"), obj.attr("code"));
                std::printf("loss2 %.9g\n", obj2.attr("loss").cast<float>());
                std::printf("code_len %d\n", pyOptimizer.attr("code_len").cast<int>());
                print_arr("t_cam_obj2", obj2.attr("t_cam_obj"));
                // mesh extractor, as CreateNewMapObjects / ProcessDetectedObjects call it: code in, .vertices cast to MatrixXf,
                // .faces cast to MatrixXi                                             (LocalMapping_util.cc:194-196,426-428)
                py::object pyMesh = pyMeshExtractor.attr("extract_mesh_from_code")(obj.attr("code"));
                py::array_t<float, py::array::c_style | py::array::forcecast> verts(pyMesh.attr("vertices"));
                py::array_t<int, py::array::c_style | py::array::forcecast> faces(pyMesh.attr("faces"));
                std::printf("mesh_shape %lld %lld %lld %lld\n", (long long)verts.shape(0), (long long)verts.shape(1), (long long)faces.shape(0),
                            (long long)faces.shape(1));
                print_arr("mesh_vertices", pyMesh.attr("vertices"));
                std::printf("mesh_faces");
                for (py::ssize_t i = 0; i < faces.size(); ++i) std::printf(" %d", faces.data()[i]);
                std::printf("\n");
                py::object grid = pyMeshExtractor.attr("decode_grid")(obj.attr("code"));
                std::printf("grid_size %lld\n", (long long)py::array(grid).size());
            }
            // missing attribute -> KeyError, as ForceKeyErrorDict does (reconstruct/utils.py:82-84)
            try { py::object missing = obj.attr("no_such_field"); (void)missing; std::printf("keyerror 0\n"); }
            catch (py::error_already_set& e) { std::printf("keyerror %d\n", e.matches(PyExc_KeyError) ? 1 : 0); }
        } catch (std::exception& e) {
            std::fprintf(stderr, "harness exception: %s\n", e.what());
            rc = 1;
        }
    });
    local_mapping.join();
    PyEval_RestoreThread(main_state);
    pyCfg = py::object();
    pyDecoder = py::object();
    py::finalize_interpreter();
    return rc;
}
