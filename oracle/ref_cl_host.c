/* TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * ref_cl_host: runs the REFERENCE'S OWN OpenCL kernel on a real OpenCL device of this machine.
 *
 * oracle/build_ref.py compiles /root/reference/Source/kernels.cl where it lies with ROCm clang for
 * amdgcn-amd-amdhsa / gfx950 — linked against AMD's own OpenCL builtin library (opencl.bc, ocml.bc, ockl.bc), with the
 * reference's own build options (skeleton.cpp:407) — into oracle/_ref/ref_<variant>_gfx950.co.  No builtin is
 * replaced by anything of ours: this is the kernel the reference's clBuildProgram would produce on this GPU.  This
 * program is what stands in for the reference's host (skeleton.cpp needs SDL2, which the image lacks): it loads that
 * code object with clCreateProgramWithBinary, passes the twelve arguments of `draw` (kernels.cl:368-371) the way
 * opencl_initialise / offload_rendering set them (skeleton.cpp:451-471, :160-167), launches one NDRange of
 * W x H work-items in groups of 128 x 4 (skeleton.cpp:28-29, :170-172), reads the ARGB frame back and writes it to a
 * file.  It also lists every OpenCL platform / device it finds (the probe SURVEY.md 8(d) asks for:
 * is there a CL_DEVICE_TYPE_CPU device on this box?) and times the kernel with OpenCL profiling events.
 *
 *   ref_cl_host probe
 *   ref_cl_host run <code-object> <job.bin> <out.bin>
 *
 * job.bin (written by oracle/ref_gpu.py), little endian:
 *   int32 n, W, H, reps;  float focal;  float rot[12];  float cam[4];  float light[4];
 *   float vertices[3n][4];  float normals[n][4];  float colors[n][4]
 * out.bin: W*H uint32 ARGB words.   stdout: one JSON line.
 */
#define CL_TARGET_OPENCL_VERSION 120
#include <CL/cl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void die(const char* what, cl_int err) {
  fprintf(stderr, "ref_cl_host: %s failed (%d)\n", what, (int)err);
  exit(2);
}
#define CK(call) do { cl_int e_ = (call); if (e_ != CL_SUCCESS) die(#call, e_); } while (0)

static void* slurp(const char* path, size_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "ref_cl_host: cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc((size_t)n + 1);
  if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "ref_cl_host: short read of %s\n", path); exit(2); }
  fclose(f);
  *len = (size_t)n;
  return p;
}

/* Lists platforms and devices; returns the first GPU device (or the first device of any kind when want_any). */
static int probe(cl_platform_id* plat_out, cl_device_id* dev_out, int print) {
  cl_platform_id plats[8];
  cl_uint np = 0;
  cl_int err = clGetPlatformIDs(8, plats, &np);
  int cpus = 0, gpus = 0, found = 0;
  if (print) printf("{\"platforms\": [");
  if (err != CL_SUCCESS) np = 0;
  for (cl_uint p = 0; p < np; ++p) {
    char pname[256] = "";
    clGetPlatformInfo(plats[p], CL_PLATFORM_NAME, sizeof pname, pname, NULL);
    cl_device_id devs[32];
    cl_uint nd = 0;
    if (clGetDeviceIDs(plats[p], CL_DEVICE_TYPE_ALL, 32, devs, &nd) != CL_SUCCESS) nd = 0;
    if (print) printf("%s{\"name\": \"%s\", \"devices\": [", p ? ", " : "", pname);
    for (cl_uint d = 0; d < nd; ++d) {
      char dname[256] = "";
      cl_device_type ty = 0;
      cl_uint cus = 0;
      clGetDeviceInfo(devs[d], CL_DEVICE_NAME, sizeof dname, dname, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_TYPE, sizeof ty, &ty, NULL);
      clGetDeviceInfo(devs[d], CL_DEVICE_MAX_COMPUTE_UNITS, sizeof cus, &cus, NULL);
      if (ty & CL_DEVICE_TYPE_CPU) ++cpus;
      if (ty & CL_DEVICE_TYPE_GPU) {
        ++gpus;
        if (!found) { *plat_out = plats[p]; *dev_out = devs[d]; found = 1; }
      }
      if (print)
        printf("%s{\"name\": \"%s\", \"type\": \"%s\", \"compute_units\": %u}", d ? ", " : "", dname,
               (ty & CL_DEVICE_TYPE_CPU) ? "cpu" : (ty & CL_DEVICE_TYPE_GPU) ? "gpu" : "other", cus);
    }
    if (print) printf("]}");
  }
  if (print) printf("], \"opencl_cpu_devices\": %d, \"opencl_gpu_devices\": %d}\n", cpus, gpus);
  return found;
}

struct job_head {
  int32_t n, W, H, reps;
  float focal, rot[12], cam[4], light[4];
};

int main(int argc, char** argv) {
  cl_platform_id plat = 0;
  cl_device_id dev = 0;
  if (argc >= 2 && !strcmp(argv[1], "probe")) {
    probe(&plat, &dev, 1);
    return 0;
  }
  if (argc != 5 || strcmp(argv[1], "run")) {
    fprintf(stderr, "usage: ref_cl_host probe | run <code-object> <job.bin> <out.bin>\n");
    return 2;
  }
  if (!probe(&plat, &dev, 0)) { fprintf(stderr, "ref_cl_host: no OpenCL GPU device\n"); return 3; }

  size_t blen = 0, jlen = 0;
  const unsigned char* bin = (const unsigned char*)slurp(argv[2], &blen);
  const char* job = (const char*)slurp(argv[3], &jlen);
  struct job_head h;
  if (jlen < sizeof h) { fprintf(stderr, "ref_cl_host: job file too short\n"); return 2; }
  memcpy(&h, job, sizeof h);
  const size_t n = (size_t)h.n;
  if (h.n < 1 || h.W < 1 || h.H < 1 || jlen != sizeof h + n * 5 * 16) { fprintf(stderr, "ref_cl_host: bad job file\n"); return 2; }
  /* the reference launches groups of 128 x 4 and has no bounds check (skeleton.cpp:170-171, kernels.cl:378-380) */
  if (h.W % 128 || h.H % 4) { fprintf(stderr, "ref_cl_host: W must be a multiple of 128 and H of 4\n"); return 2; }
  const float* verts = (const float*)(job + sizeof h);
  const float* normals = verts + n * 12;
  const float* colors = normals + n * 4;

  cl_int err;
  cl_context ctx = clCreateContext(NULL, 1, &dev, NULL, NULL, &err);
  if (err != CL_SUCCESS) die("clCreateContext", err);
  cl_command_queue q = clCreateCommandQueue(ctx, dev, CL_QUEUE_PROFILING_ENABLE, &err);
  if (err != CL_SUCCESS) die("clCreateCommandQueue", err);
  cl_int bstat = 0;
  cl_program prog = clCreateProgramWithBinary(ctx, 1, &dev, &blen, &bin, &bstat, &err);
  if (err != CL_SUCCESS) die("clCreateProgramWithBinary", err);
  err = clBuildProgram(prog, 1, &dev, "", NULL, NULL);
  if (err != CL_SUCCESS) {
    char log[4096] = "";
    clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, sizeof log, log, NULL);
    fprintf(stderr, "build log: %s\n", log);
    die("clBuildProgram", err);
  }
  cl_kernel k = clCreateKernel(prog, "draw", &err);
  if (err != CL_SUCCESS) die("clCreateKernel(draw)", err);

  const size_t px = (size_t)h.W * (size_t)h.H;
  cl_mem d_screen = clCreateBuffer(ctx, CL_MEM_WRITE_ONLY, px * 4, NULL, &err);
  if (err != CL_SUCCESS) die("clCreateBuffer(screen)", err);
  cl_mem d_verts = clCreateBuffer(ctx, CL_MEM_READ_ONLY, n * 48, NULL, &err);
  if (err != CL_SUCCESS) die("clCreateBuffer(vertices)", err);
  cl_mem d_normals = clCreateBuffer(ctx, CL_MEM_READ_ONLY, n * 16, NULL, &err);
  if (err != CL_SUCCESS) die("clCreateBuffer(normals)", err);
  cl_mem d_colors = clCreateBuffer(ctx, CL_MEM_READ_ONLY, n * 16, NULL, &err);
  if (err != CL_SUCCESS) die("clCreateBuffer(colors)", err);
  cl_mem d_rot = clCreateBuffer(ctx, CL_MEM_READ_ONLY, 48, NULL, &err);
  if (err != CL_SUCCESS) die("clCreateBuffer(rot)", err);
  CK(clEnqueueWriteBuffer(q, d_verts, CL_TRUE, 0, n * 48, verts, 0, NULL, NULL));
  CK(clEnqueueWriteBuffer(q, d_normals, CL_TRUE, 0, n * 16, normals, 0, NULL, NULL));
  CK(clEnqueueWriteBuffer(q, d_colors, CL_TRUE, 0, n * 16, colors, 0, NULL, NULL));
  CK(clEnqueueWriteBuffer(q, d_rot, CL_TRUE, 0, 48, h.rot, 0, NULL, NULL));

  cl_int ni = h.n;
  CK(clSetKernelArg(k, 0, sizeof(cl_mem), &d_screen));
  CK(clSetKernelArg(k, 1, sizeof(cl_mem), &d_verts));
  CK(clSetKernelArg(k, 2, sizeof(cl_mem), &d_normals));
  CK(clSetKernelArg(k, 3, sizeof(cl_mem), &d_colors));
  CK(clSetKernelArg(k, 4, sizeof(cl_mem), &d_rot));
  CK(clSetKernelArg(k, 5, 16, h.cam));          /* float3 by value = 16 bytes */
  CK(clSetKernelArg(k, 6, 16, h.light));
  CK(clSetKernelArg(k, 7, sizeof(cl_int), &ni));
  CK(clSetKernelArg(k, 8, sizeof(cl_float), &h.focal));
  CK(clSetKernelArg(k, 9, n * 48, NULL));       /* __local scratch, sized as skeleton.cpp:466-471 */
  CK(clSetKernelArg(k, 10, n * 16, NULL));
  CK(clSetKernelArg(k, 11, n * 16, NULL));

  /* Work-group shape: 128 x 4 as the reference's host asks (skeleton.cpp:28-29) where the device allows it.  AMD's OpenCL
   * compiles kernels for at most 256 work-items per group unless told otherwise (CL_KERNEL_WORK_GROUP_SIZE; the
   * reference's 512 is refused with CL_INVALID_WORK_GROUP_SIZE), so the group is halved in y until it fits.  No value
   * depends on the shape: work-items are independent and every group stages the whole scene (kernels.cl:374-380). */
  size_t local[2] = {128, 4}, kwg = 0;
  CK(clGetKernelWorkGroupInfo(k, dev, CL_KERNEL_WORK_GROUP_SIZE, sizeof kwg, &kwg, NULL));
  while (local[0] * local[1] > kwg && local[1] > 1) local[1] /= 2;
  while (local[0] * local[1] > kwg && local[0] > 1) local[0] /= 2;
  const size_t global[2] = {(size_t)h.W, (size_t)h.H};
  const int reps = h.reps > 0 ? h.reps : 1;
  double sum_ms = 0.0, min_ms = 1e30;
  for (int r = 0; r < reps + 1; ++r) {          /* launch 0 is a warm-up */
    cl_event ev;
    CK(clEnqueueNDRangeKernel(q, k, 2, NULL, global, local, 0, NULL, &ev));
    CK(clWaitForEvents(1, &ev));
    cl_ulong t0 = 0, t1 = 0;
    CK(clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_START, sizeof t0, &t0, NULL));
    CK(clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_END, sizeof t1, &t1, NULL));
    clReleaseEvent(ev);
    const double ms = (double)(t1 - t0) * 1e-6;
    if (r > 0 || reps == 0) { sum_ms += ms; if (ms < min_ms) min_ms = ms; }
  }
  uint32_t* out = (uint32_t*)malloc(px * 4);
  CK(clEnqueueReadBuffer(q, d_screen, CL_TRUE, 0, px * 4, out, 0, NULL, NULL));
  FILE* f = fopen(argv[4], "wb");
  if (!f || fwrite(out, 4, px, f) != px) { fprintf(stderr, "ref_cl_host: cannot write %s\n", argv[4]); return 2; }
  fclose(f);

  char dname[256] = "";
  clGetDeviceInfo(dev, CL_DEVICE_NAME, sizeof dname, dname, NULL);
  printf("{\"device\": \"%s\", \"W\": %d, \"H\": %d, \"n\": %d, \"local\": [%d, %d], \"kernel_max_work_group\": %d, \"launches\": %d, \"kernel_ms_mean\": %.6f, \"kernel_ms_min\": %.6f}\n",
         dname, h.W, h.H, h.n, (int)local[0], (int)local[1], (int)kwg, reps, sum_ms / reps, min_ms);
  clReleaseMemObject(d_screen); clReleaseMemObject(d_verts); clReleaseMemObject(d_normals);
  clReleaseMemObject(d_colors); clReleaseMemObject(d_rot);
  clReleaseKernel(k); clReleaseProgram(prog); clReleaseCommandQueue(q); clReleaseContext(ctx);
  return 0;
}
