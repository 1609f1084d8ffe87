// C++ embed harness: drives the Python mirror exactly the way DSP-SLAM's C++ does, without Eigen/OpenCV.
//
// Replays, from a NON-MAIN std::thread under PyGILState_Ensure (reference include/System.h:56-70), the call / cast
// sequence of   src/System.cc:90-99,152 (interpreter, sys.path, get_configs, get_decoder, GIL release),
//               src/LocalMapping.cc:38-40 (Optimizer / MeshExtractor construction),
//               src/LocalMapping_util.cc:109-110 (estimate_pose_cam_obj -> Matrix4f),
//               src/LocalMapping_util.cc:179-191 (reconstruct_object -> is_good / t_cam_obj / code),
//               src/LocalMapping_util.cc:391-413 (5-argument form with a warm-start code, loss, code_len),
//               src/LocalMapping_util.cc:194-196,426-428 (extract_mesh_from_code -> vertices MatrixXf / faces MatrixXi).
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
    if (argc < 4) { std::fprintf(stderr, "usage: embed_harness <mirror_dir> <cfg.json> <inputs.npz>\n"); return 2; }
    const std::string mirror = argv[1], cfg_file = argv[2], npz = argv[3];
    std::setvbuf(stdout, nullptr, _IOLBF, 1 << 16);                            // whole lines, so that Python's own prints cannot land inside one
    py::initialize_interpreter();                                              // System.cc:90
    py::object pyCfg, pyDecoder;
    {
        py::module::import("sys").attr("path").attr("insert")(0, mirror);      // System.cc:92 appends "./"; the mirror goes first
        py::module io_utils = py::module::import("reconstruct.utils");         // System.cc:94
        pyCfg = io_utils.attr("get_configs")(cfg_file);                        // System.cc:96
        pyDecoder = io_utils.attr("get_decoder")(pyCfg);                       // System.cc:97
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
