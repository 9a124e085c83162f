/*
 * pe_run.c -- TEST INFRASTRUCTURE (oracle side), OPT-IN: run functions of the reference's SHIPPED x64 binaries in place, confined.
 *
 * The reference cannot be built in this image, but it ships the x86-64 machine code it runs
 * (/root/reference/stitching/OpenCV2.4.11-Stitching-64bit/x64/Release/opencv_{core,imgproc,stitching}2411.dll and the project's exe).
 * This maps PE32+ images into the CALLING process (tests/ref_run.py: a short-lived child python, never the test runner itself), binds
 * their imports, and calls functions by address with the Microsoft x64 calling convention.  Nothing of the reference is copied or
 * rebuilt: no headers, no libraries.
 *
 * Safety ("high"): the images are opaque machine code from an untrusted tree, so
 *   - nothing here is built or run by the default build / test path: tests/ref_run.py refuses unless MST_RUN_REFERENCE_BINARIES=1;
 *   - the caller hands over the image BYTES after checking their SHA-256 against the pinned values (tests/ref_run.py: PINNED);
 *   - before the first call into an image the process must pe_lockdown(): it drops to uid/gid 65534 when it was root, moves to an empty
 *     scratch directory, sets no_new_privs and installs a seccomp filter on ALL threads that leaves only memory management, reads and
 *     writes on descriptors that are already open, clocks and exit -- no open, no socket, no exec, no fork/clone, no kill, no ptrace;
 *     pe_call() refuses to run before that.  What the code can then do is compute, fault, or write to the pipe / output file its parent
 *     gave it (the parent parses that as plain arrays, allow_pickle off).
 *
 * Two binding policies:
 *   PE_LEAF (0)  one image, EVERY import trapped except cv::checkHardwareSupport and memset / memcpy / memmove: import-free leaf
 *                functions only (round 5's claim, unchanged);
 *   PE_FULL (1)  several images mapped together (relocated where two want the same base), imports bound to the other images' exports
 *                by name, and the loader / C-runtime SERVICES answered by the host: allocation (malloc family, operator new / delete),
 *                byte movers, locks and TLS slots (single thread: no-ops and an array), the CRT's start-up plumbing (_initterm, atexit
 *                registration, pointer encoding), clocks (constants).  None of these computes a pixel.  The two math imports the
 *                blender's prepare() needs -- log and ceil, for `ceil(log(max_len) / log(2.0))`, the band-count clamp -- are answered by
 *                the host libm and every such call is LOGGED (pe_math_log) so that the caller can check the clamp was nowhere near an
 *                integer boundary.  Also answered: sqrt (correctly rounded by definition), std::numeric_limits<T> members (constants), and --
 *                only after pe_allow_libm(1), the warp-map jobs -- the projectors' float math library (sinf, cosf, atan2f ...), counted.
 *                Every other import (math, I/O, locale, the other OpenCV modules) stays trapped by name.
 *
 * Build: gcc -O1 -shared -fPIC -o libpe_run.so pe_run.c -lm
 */
#define _GNU_SOURCE
#include <errno.h>
#include <grp.h>
#include <math.h>
#include <signal.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>
#include <linux/audit.h>
#include <linux/filter.h>
#include <linux/seccomp.h>

#ifndef MAP_FIXED_NOREPLACE
#define MAP_FIXED_NOREPLACE 0x100000
#endif
#define MS __attribute__((ms_abi))

enum { PE_LEAF = 0, PE_FULL = 1 };
#define MAX_IMAGES 4
#define MAX_STUBS 4096

typedef struct {
    uint8_t *img;
    uint64_t base, preferred;
    uint32_t size, entry;
    char name[64];
} image_t;

static image_t g_img[MAX_IMAGES];
static int g_n_img;
static int g_locked;
static int g_no_sse;
static int g_verbose;

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

static void say(const char *s)
{
    ssize_t r = write(2, s, strlen(s));
    (void)r;
}

/* which image an address lies in, as "name+rva" */
static const char *where(uint64_t a, char *buf, size_t n)
{
    for (int i = 0; i < g_n_img; ++i)
        if (a >= g_img[i].base && a < g_img[i].base + g_img[i].size) {
            snprintf(buf, n, "%s+0x%llx (preferred 0x%llx)", g_img[i].name, (unsigned long long)(a - g_img[i].base),
                     (unsigned long long)(a - g_img[i].base + g_img[i].preferred));
            return buf;
        }
    snprintf(buf, n, "%p", (void *)a);
    return buf;
}

/* ---- named traps: one stub per unanswered import, so that a call says WHICH import the path needs ------------------------------ */
static uint8_t *g_stubs;
static int g_n_stubs;
static char g_stub_name[MAX_STUBS][96];

void trap_named(uint64_t idx, uint64_t ret)
{
    char msg[400], w[160];
    int n = snprintf(msg, sizeof msg, "pe_run: the reference code called the unanswered IMPORT %s (from %s) -- aborting\n",
                     idx < (uint64_t)g_n_stubs ? g_stub_name[idx] : "?", where(ret, w, sizeof w));
    ssize_t r = write(2, msg, (size_t)n);
    (void)r;
    _exit(77);
}

__attribute__((naked)) static void trap_common(void)
{
    __asm__ volatile("mov %r10, %rdi\n\tmov (%rsp), %rsi\n\tand $-16, %rsp\n\tcall trap_named\n\thlt\n");
}

static uint64_t make_stub(const char *dll, const char *fn)
{
    if (!g_stubs) {
        g_stubs = (uint8_t *)mmap(NULL, MAX_STUBS * 32, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (g_stubs == MAP_FAILED) { g_stubs = NULL; return 0; }
    }
    if (g_n_stubs >= MAX_STUBS) return (uint64_t)(uintptr_t)g_stubs;       /* (the first stub: still a trap) */
    const int i = g_n_stubs++;
    snprintf(g_stub_name[i], sizeof g_stub_name[i], "%s!%s", dll, fn);
    uint8_t *s = g_stubs + 32 * i;
    uint64_t v = (uint64_t)i, t = (uint64_t)(uintptr_t)&trap_common;
    s[0] = 0x49; s[1] = 0xBA; memcpy(s + 2, &v, 8);                        /* mov r10, imm64 */
    s[10] = 0x48; s[11] = 0xB8; memcpy(s + 12, &t, 8);                     /* mov rax, imm64 */
    s[20] = 0xFF; s[21] = 0xE0;                                            /* jmp rax */
    return (uint64_t)(uintptr_t)s;
}

/* a fault inside the mapped code: say where (instruction and address), then exit 78 */
static void on_fault(int sig, siginfo_t *si, void *uc_)
{
    ucontext_t *uc = (ucontext_t *)uc_;
    char msg[400], w[160];
    int n = snprintf(msg, sizeof msg, "pe_run: signal %d at rip %s touching %p (rsp %p)\n", sig,
                     where((uint64_t)uc->uc_mcontext.gregs[REG_RIP], w, sizeof w), si->si_addr, (void *)uc->uc_mcontext.gregs[REG_RSP]);
    ssize_t r = write(2, msg, (size_t)n);
    (void)r;
    _exit(78);
}

/* ---- answered imports --------------------------------------------------------------------------------------------------------- */
/* PE_LEAF: cv::checkHardwareSupport(int feature) of the reference's own opencv_core2411.dll, which the imgproc loops ask before taking
 * their SSE bodies.  The real function returns what cpuid said when the DLL was loaded; here it answers as every x86-64 CPU does -- MMX
 * (1), SSE (2), SSE2 (3): yes; anything newer: no.  PE_RUN_NO_SSE=1 answers "no" to everything instead (the scalar loops alone).  Under
 * PE_FULL the import is bound to core's own function. */
static uint64_t MS check_hardware_support(int feature) { return (!g_no_sse && feature >= 1 && feature <= 3) ? 1u : 0u; }

/* byte movers (both policies): no arithmetic, nothing a restatement could get wrong */
static void *MS crt_memset(void *d, int c, size_t n) { return memset(d, c, n); }
static void *MS crt_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
static void *MS crt_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }
static int MS crt_memcmp(const void *a, const void *b, size_t n) { return memcmp(a, b, n); }
static size_t MS crt_strlen(const char *s) { return strlen(s); }

/* allocation (PE_FULL): the host allocator */
static void *MS crt_malloc(size_t n) { return malloc(n ? n : 1); }
static void MS crt_free(void *p) { free(p); }
static void *MS crt_calloc(size_t a, size_t b) { return calloc(a ? a : 1, b ? b : 1); }
static void *MS crt_realloc(void *p, size_t n) { return realloc(p, n ? n : 1); }
static void *MS crt_aligned_malloc(size_t n, size_t al)
{
    void *p = NULL;
    return posix_memalign(&p, al < sizeof(void *) ? sizeof(void *) : al, n ? n : 1) == 0 ? p : NULL;
}

/* C++ exceptions cannot unwind here (no SEH): a throw ends the run.  cv::Exception is {vptr, std::string msg, int code, ...}: print the
 * MSVC std::string at +8 ({union {char buf[16]; char* ptr;}, size_t size, size_t res}) when it looks like one. */
static void MS crt_throw(void *obj, void *info)
{
    char msg[700];
    int n = snprintf(msg, sizeof msg, "pe_run: the reference code THREW a C++ exception (object %p) -- aborting\n", obj);
    ssize_t r = write(2, msg, (size_t)n);
    if (obj) {
        const uint8_t *s = (const uint8_t *)obj + 8;
        const uint64_t size = rd64(s + 16), res = rd64(s + 24);
        if (size < 600 && res >= size && res < (1u << 20)) {
            const char *txt = res >= 16 ? (const char *)(uintptr_t)rd64(s) : (const char *)s;
            r = write(2, "  what: ", 8);
            r = write(2, txt, size);
            r = write(2, "\n", 1);
        }
    }
    (void)r; (void)info;
    _exit(79);
}

/* the CRT's start-up plumbing, as the real _DllMainCRTStartup uses it */
typedef void (MS *pvfv)(void);
typedef int (MS *pifv)(void);
static void MS crt_initterm(pvfv *a, pvfv *z) { for (; a < z; ++a) if (*a) (*a)(); }
static int MS crt_initterm_e(pifv *a, pifv *z) { int r = 0; for (; a < z && !r; ++a) if (*a) r = (*a)(); return r; }
static void *MS crt_identity(void *p) { return p; }
static void *MS crt_null(void) { return NULL; }
static uint64_t MS crt_zero(void) { return 0; }
static uint64_t MS crt_one(void) { return 1; }
static void MS crt_nop(void) {}
/* atexit registration: accepted, never run (the process exits without running the images' destructors) */
static void *MS crt_onexit(void *fn) { return fn; }
static void *MS crt_dllonexit(void *fn, void *a, void *b) { (void)a; (void)b; return fn; }

/* KERNEL32: one thread, so locks are no-ops and TLS is an array; clocks are constants (nothing on the pixel path reads them) */
static void *g_tls[64];
static int g_tls_n;
static uint32_t MS k32_tls_alloc(void) { return g_tls_n < 64 ? (uint32_t)g_tls_n++ : 0xFFFFFFFFu; }
static void *MS k32_tls_get(uint32_t i) { return i < 64 ? g_tls[i] : NULL; }
static int MS k32_tls_set(uint32_t i, void *v) { if (i < 64) { g_tls[i] = v; return 1; } return 0; }
static int MS k32_qpc(int64_t *v) { static int64_t t = 1000; *v = (t += 1000); return 1; }
static int MS k32_qpf(int64_t *v) { *v = 1000000; return 1; }
static void MS k32_filetime(uint64_t *v) { *v = 0x01D00000DEADBEEFull; }
static void MS k32_sysinfo(uint8_t *si) { memset(si, 0, 48); uint32_t one = 1, page = 4096; memcpy(si + 4, &page, 4); memcpy(si + 32, &one, 4); }

/* the two math imports of MultiBandBlender::prepare's band-count clamp: answered by the host libm, every call logged */
#define MATH_LOG_MAX 64
static struct { char fn[8]; double x, y; } g_math[MATH_LOG_MAX];
static int g_n_math;
static void math_note(const char *fn, double x, double y)
{
    if (g_n_math < MATH_LOG_MAX) { snprintf(g_math[g_n_math].fn, 8, "%s", fn); g_math[g_n_math].x = x; g_math[g_n_math].y = y; }
    ++g_n_math;
}
static double MS crt_log(double x) { double y = log(x); math_note("log", x, y); return y; }
static double MS crt_ceil(double x) { double y = ceil(x); math_note("ceil", x, y); return y; }
/* sin / cos: imgproc's static initialiser builds ALL interpolation tables when the image loads (imgwarp.cpp: `doInitAllInterTab2D`), the
 * Lanczos4 one through sin / cos.  Answered by the host libm ONLY while the entry points run (phase 0) -- no path run here reads the
 * Lanczos table (the bilinear one is exact arithmetic: products of small integers over 32) -- and trapped afterwards. */
static int g_phase, g_n_trig;
static double MS crt_sin(double x)
{
    if (g_phase) { say("pe_run: the reference code called sin() outside the images' initialisers -- aborting\n"); _exit(77); }
    ++g_n_trig;
    return sin(x);
}
static double MS crt_cos(double x)
{
    if (g_phase) { say("pe_run: the reference code called cos() outside the images' initialisers -- aborting\n"); _exit(77); }
    ++g_n_trig;
    return cos(x);
}
int pe_phase(int phase) { g_phase = phase; return g_n_trig; }

/* sqrt (GainCompensator::feed's intensity sums, should the build call the CRT instead of inlining sqrtsd): IEEE 754 requires the correctly
 * rounded result -- every conforming sqrt returns the same double.  std::numeric_limits<double>::epsilon() (cv::LU's singularity test) is the
 * constant 2^-52. */
static double MS crt_sqrt(double x) { return sqrt(x); }
static double MS msvcp_dbl_epsilon(void) { return 2.220446049250313e-16; }
static float MS msvcp_flt_epsilon(void) { return 1.1920929e-7f; }
/* the other std::numeric_limits members MSVCP100 exports as functions: constants of the type, nothing computed */
static int MS msvcp_int_max(void) { return 2147483647; }
static int MS msvcp_int_min(void) { return -2147483647 - 1; }
static float MS msvcp_flt_max(void) { return 3.402823466e+38f; }
static float MS msvcp_flt_min(void) { return 1.175494351e-38f; }
static double MS msvcp_dbl_max(void) { return 1.7976931348623158e+308; }
static double MS msvcp_dbl_min(void) { return 2.2250738585072014e-308; }
static float MS msvcp_flt_inf(void) { return __builtin_inff(); }
static double MS msvcp_dbl_inf(void) { return __builtin_inf(); }

/* The float math library behind the PROJECTORS (sinf, cosf, atan2f ... of MSVCR100, which the tree does not ship).  Answered by the host's
 * libm ONLY after pe_allow_libm(1) -- the warp-map jobs, which say so in their output: such a run pins the projector's formulas, the order of
 * its float operations, the ROI detection and the map layout against the oracle UNDER THE SAME libm; it cannot pin MSVCR100's function values. */
static int g_allow_libm;
static long g_n_libm;
int pe_allow_libm(int on) { g_allow_libm = on; return (int)g_n_libm; }
static void libm_gate(const char *fn)
{
    if (!g_allow_libm) { say("pe_run: the reference code called the float math library ("); say(fn); say(") in a run that did not allow it -- aborting\n"); _exit(77); }
    ++g_n_libm;
}
#define LIBM1(name) static float MS crt_##name(float x) { libm_gate(#name); return name(x); }
LIBM1(sinf) LIBM1(cosf) LIBM1(tanf) LIBM1(asinf) LIBM1(acosf) LIBM1(atanf) LIBM1(sinhf) LIBM1(coshf) LIBM1(logf) LIBM1(expf) LIBM1(sqrtf)
static float MS crt_atan2f(float y, float x) { libm_gate("atan2f"); return atan2f(y, x); }
static double MS crt_pow(double x, double y) { libm_gate("pow"); return pow(x, y); }

int pe_math_log(int i, char *fn, double *x, double *y)
{
    if (i < 0 || i >= g_n_math || i >= MATH_LOG_MAX) return g_n_math;
    memcpy(fn, g_math[i].fn, 8); *x = g_math[i].x; *y = g_math[i].y;
    return g_n_math;
}

typedef struct { const char *name; void *fn; int policy; } shim_t;   /* policy: the lowest policy that answers it */
static const shim_t SHIMS[] = {
    {"memset", (void *)crt_memset, PE_LEAF}, {"memcpy", (void *)crt_memcpy, PE_LEAF}, {"memmove", (void *)crt_memmove, PE_LEAF},
    {"memcmp", (void *)crt_memcmp, PE_FULL}, {"strlen", (void *)crt_strlen, PE_FULL},
    {"malloc", (void *)crt_malloc, PE_FULL}, {"free", (void *)crt_free, PE_FULL}, {"calloc", (void *)crt_calloc, PE_FULL},
    {"realloc", (void *)crt_realloc, PE_FULL}, {"_aligned_malloc", (void *)crt_aligned_malloc, PE_FULL}, {"_aligned_free", (void *)crt_free, PE_FULL},
    {"??2@YAPEAX_K@Z", (void *)crt_malloc, PE_FULL}, {"??3@YAXPEAX@Z", (void *)crt_free, PE_FULL},           /* operator new / delete */
    {"??_U@YAPEAX_K@Z", (void *)crt_malloc, PE_FULL}, {"??_V@YAXPEAX@Z", (void *)crt_free, PE_FULL},          /* operator new[] / delete[] */
    {"_malloc_crt", (void *)crt_malloc, PE_FULL},
    {"_CxxThrowException", (void *)crt_throw, PE_FULL},
    {"_initterm", (void *)crt_initterm, PE_FULL}, {"_initterm_e", (void *)crt_initterm_e, PE_FULL},
    {"_onexit", (void *)crt_onexit, PE_FULL}, {"__dllonexit", (void *)crt_dllonexit, PE_FULL},
    {"_lock", (void *)crt_nop, PE_FULL}, {"_unlock", (void *)crt_nop, PE_FULL},
    {"_encode_pointer", (void *)crt_identity, PE_FULL}, {"_decode_pointer", (void *)crt_identity, PE_FULL}, {"_encoded_null", (void *)crt_null, PE_FULL},
    {"__clean_type_info_names_internal", (void *)crt_nop, PE_FULL}, {"_crt_debugger_hook", (void *)crt_nop, PE_FULL},
    {"log", (void *)crt_log, PE_FULL}, {"ceil", (void *)crt_ceil, PE_FULL}, {"sin", (void *)crt_sin, PE_FULL}, {"cos", (void *)crt_cos, PE_FULL},
    {"sinf", (void *)crt_sinf, PE_FULL}, {"cosf", (void *)crt_cosf, PE_FULL}, {"tanf", (void *)crt_tanf, PE_FULL}, {"asinf", (void *)crt_asinf, PE_FULL},
    {"acosf", (void *)crt_acosf, PE_FULL}, {"atanf", (void *)crt_atanf, PE_FULL}, {"atan2f", (void *)crt_atan2f, PE_FULL}, {"sinhf", (void *)crt_sinhf, PE_FULL},
    {"coshf", (void *)crt_coshf, PE_FULL}, {"logf", (void *)crt_logf, PE_FULL}, {"expf", (void *)crt_expf, PE_FULL}, {"sqrtf", (void *)crt_sqrtf, PE_FULL},
    {"pow", (void *)crt_pow, PE_FULL},
    {"sqrt", (void *)crt_sqrt, PE_FULL}, {"?epsilon@?$numeric_limits@N@std@@SANXZ", (void *)msvcp_dbl_epsilon, PE_FULL},
    {"?epsilon@?$numeric_limits@M@std@@SAMXZ", (void *)msvcp_flt_epsilon, PE_FULL},
    {"?max@?$numeric_limits@H@std@@SAHXZ", (void *)msvcp_int_max, PE_FULL}, {"?min@?$numeric_limits@H@std@@SAHXZ", (void *)msvcp_int_min, PE_FULL},
    {"?max@?$numeric_limits@M@std@@SAMXZ", (void *)msvcp_flt_max, PE_FULL}, {"?min@?$numeric_limits@M@std@@SAMXZ", (void *)msvcp_flt_min, PE_FULL},
    {"?max@?$numeric_limits@N@std@@SANXZ", (void *)msvcp_dbl_max, PE_FULL}, {"?min@?$numeric_limits@N@std@@SANXZ", (void *)msvcp_dbl_min, PE_FULL},
    {"?infinity@?$numeric_limits@M@std@@SAMXZ", (void *)msvcp_flt_inf, PE_FULL}, {"?infinity@?$numeric_limits@N@std@@SANXZ", (void *)msvcp_dbl_inf, PE_FULL},
    {"EncodePointer", (void *)crt_identity, PE_FULL}, {"DecodePointer", (void *)crt_identity, PE_FULL},
    {"InitializeCriticalSection", (void *)crt_nop, PE_FULL}, {"InitializeCriticalSectionAndSpinCount", (void *)crt_one, PE_FULL},
    {"DeleteCriticalSection", (void *)crt_nop, PE_FULL}, {"EnterCriticalSection", (void *)crt_nop, PE_FULL},
    {"LeaveCriticalSection", (void *)crt_nop, PE_FULL}, {"TryEnterCriticalSection", (void *)crt_one, PE_FULL},
    {"TlsAlloc", (void *)k32_tls_alloc, PE_FULL}, {"TlsGetValue", (void *)k32_tls_get, PE_FULL}, {"TlsSetValue", (void *)k32_tls_set, PE_FULL},
    {"QueryPerformanceCounter", (void *)k32_qpc, PE_FULL}, {"QueryPerformanceFrequency", (void *)k32_qpf, PE_FULL},
    {"GetSystemTimeAsFileTime", (void *)k32_filetime, PE_FULL}, {"GetCurrentProcessId", (void *)crt_one, PE_FULL},
    {"GetCurrentThreadId", (void *)crt_one, PE_FULL}, {"GetTickCount", (void *)crt_one, PE_FULL},
    {"DisableThreadLibraryCalls", (void *)crt_one, PE_FULL}, {"IsDebuggerPresent", (void *)crt_zero, PE_FULL},
    {"GetNativeSystemInfo", (void *)k32_sysinfo, PE_FULL}, {"GetSystemInfo", (void *)k32_sysinfo, PE_FULL}, {"Sleep", (void *)crt_nop, PE_FULL},
    /* std::_Lockit (iterator-debugging / locale lock objects constructed around container operations): no-ops on one thread */
    /* stitching's static initialiser registers the nonfree feature detectors (matchers.cpp); that module is not mapped: "done" */
    {"?initModule_nonfree@cv@@YA_NXZ", (void *)crt_one, PE_FULL},
    {"?_Orphan_all@_Container_base0@std@@QEAAXXZ", (void *)crt_nop, PE_FULL},       /* release-build containers: empty bodies in MSVCP100 */
    {"?_Swap_all@_Container_base0@std@@QEAAXAEAU12@@Z", (void *)crt_nop, PE_FULL},
    {"??0_Lockit@std@@QEAA@H@Z", (void *)crt_identity, PE_FULL}, {"??1_Lockit@std@@QEAA@XZ", (void *)crt_nop, PE_FULL},
};

/* ---- mapping ---------------------------------------------------------------------------------------------------------------- */
static const uint8_t *opt_header(const uint8_t *img) { return img + rd32(img + 0x3C) + 24; }

/* Maps the PE32+ image held in `bytes` at `at` (0 = its preferred base), applying its base relocations when the two differ.
 * Returns the image's index (>= 0) or -1. */
int pe_map(const uint8_t *file, long len, uint64_t at, const char *name)
{
    if (g_locked || g_n_img >= MAX_IMAGES) return -1;
    if (len < 0x200 || rd16(file) != 0x5A4D) return -1;
    const uint8_t *nt = file + rd32(file + 0x3C);
    if (rd32(nt) != 0x00004550 || rd16(nt + 4) != 0x8664) return -1;                             /* "PE\0\0", AMD64 */
    const int n_sections = rd16(nt + 6);
    const int opt_size = rd16(nt + 20);
    const uint8_t *opt = nt + 24;
    if (rd16(opt) != 0x20B) return -1;                                                           /* PE32+ */
    const uint64_t preferred = rd64(opt + 24);
    const uint64_t base = at ? at : preferred;
    const uint32_t image_size = rd32(opt + 56), headers_size = rd32(opt + 60);
    uint8_t *img = (uint8_t *)mmap((void *)base, image_size, PROT_READ | PROT_WRITE | PROT_EXEC,
                                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
    if (img == MAP_FAILED || (uint64_t)(uintptr_t)img != base) return -1;
    memcpy(img, file, headers_size);
    const uint8_t *sec = opt + opt_size;
    for (int i = 0; i < n_sections; ++i, sec += 40) {
        const uint32_t vsize = rd32(sec + 8), va = rd32(sec + 12), raw = rd32(sec + 16), off = rd32(sec + 20);
        const uint32_t n = raw < vsize ? raw : vsize;
        if ((uint64_t)va + n <= image_size && (long)off + n <= len) memcpy(img + va, file + off, n);
    }
    if (base != preferred) {                                                                     /* IMAGE_DIRECTORY_ENTRY_BASERELOC (5) */
        const uint32_t rva = rd32(opt + 112 + 8 * 5), size = rd32(opt + 112 + 8 * 5 + 4);
        if (!rva || !size) { munmap(img, image_size); return -1; }
        const uint64_t delta = base - preferred;
        for (uint32_t o = 0; o + 8 <= size;) {
            const uint32_t page = rd32(img + rva + o), block = rd32(img + rva + o + 4);
            if (block < 8) break;
            for (uint32_t k = 8; k + 2 <= block; k += 2) {
                const uint16_t e = rd16(img + rva + o + k);
                if ((e >> 12) == 10) {                                                          /* IMAGE_REL_BASED_DIR64 */
                    uint8_t *p = img + page + (e & 0xFFF);
                    uint64_t v = rd64(p) + delta;
                    memcpy(p, &v, 8);
                } else if ((e >> 12) != 0) { munmap(img, image_size); return -1; }              /* (ABSOLUTE padding is the only other kind in x64 images) */
            }
            o += block;
        }
    }
    image_t *im = &g_img[g_n_img];
    im->img = img; im->base = base; im->preferred = preferred; im->size = image_size; im->entry = rd32(opt + 16);
    snprintf(im->name, sizeof im->name, "%s", name);
    return g_n_img++;
}

uint64_t pe_base(int idx) { return idx >= 0 && idx < g_n_img ? g_img[idx].base : 0; }
uint64_t pe_entry(int idx) { return idx >= 0 && idx < g_n_img && g_img[idx].entry ? g_img[idx].base + g_img[idx].entry : 0; }

/* the address of an exported symbol (decorated name), 0 if the image does not export it */
uint64_t pe_export(int idx, const char *name)
{
    if (idx < 0 || idx >= g_n_img) return 0;
    const image_t *im = &g_img[idx];
    const uint8_t *opt = opt_header(im->img);
    const uint32_t rva = rd32(opt + 112), size = rd32(opt + 112 + 4);
    if (!rva) return 0;
    const uint8_t *d = im->img + rva;
    const uint32_t n_names = rd32(d + 24), fns = rd32(d + 28), names = rd32(d + 32), ords = rd32(d + 36);
    for (uint32_t i = 0; i < n_names; ++i)
        if (strcmp((const char *)(im->img + rd32(im->img + names + 4 * i)), name) == 0) {
            const uint32_t f = rd32(im->img + fns + 4 * rd16(im->img + ords + 2 * i));
            if (f >= rva && f < rva + size) return 0;                                            /* a forwarder: not followed */
            return im->base + f;
        }
    return 0;
}

static int image_named(const char *dll)
{
    for (int i = 0; i < g_n_img; ++i)
        if (strcasecmp(g_img[i].name, dll) == 0) return i;
    return -1;
}

/* Binds image idx's imports under `policy`; returns the number of slots left on a trap, *answered / *bound the others. */
int pe_bind(int idx, int policy, int *n_bound, int *n_answered)
{
    if (g_locked || idx < 0 || idx >= g_n_img) return -1;
    image_t *im = &g_img[idx];
    uint8_t *img = im->img;
    const uint8_t *opt = opt_header(img);
    int trapped = 0, bound = 0, answered = 0;
    const uint32_t imp_rva = rd32(opt + 112 + 8 * 1), imp_size = rd32(opt + 112 + 8 * 1 + 4);
    if (imp_rva && imp_size)
        for (const uint8_t *d = img + imp_rva; rd32(d + 12) != 0; d += 20) {                    /* IMAGE_IMPORT_DESCRIPTOR, Name != 0 */
            const char *dll = (const char *)(img + rd32(d + 12));
            uint64_t *iat = (uint64_t *)(img + rd32(d + 16));
            const uint64_t *names = (const uint64_t *)(img + (rd32(d) ? rd32(d) : rd32(d + 16)));
            const int other = policy >= PE_FULL ? image_named(dll) : -1;
            for (int k = 0; names[k] != 0; ++k) {
                const char *nm = (names[k] >> 63) ? "(ordinal)" : (const char *)(img + (uint32_t)names[k] + 2);
                uint64_t to = 0;
                if (other >= 0 && (to = pe_export(other, nm)) != 0) ++bound;
                if (!to && other < 0) {
                    if (policy == PE_LEAF && strcmp(nm, "?checkHardwareSupport@cv@@YA_NH@Z") == 0) to = (uint64_t)(uintptr_t)&check_hardware_support;
                    for (size_t s = 0; !to && s < sizeof SHIMS / sizeof SHIMS[0]; ++s)
                        if (SHIMS[s].policy <= policy && strcmp(SHIMS[s].name, nm) == 0) to = (uint64_t)(uintptr_t)SHIMS[s].fn;
                    if (to) ++answered;
                }
                if (!to) { to = make_stub(dll, nm); ++trapped; }
                iat[k] = to;
            }
        }
    const uint32_t dimp_rva = rd32(opt + 112 + 8 * 13);                                        /* delay imports: their IAT too */
    if (dimp_rva)
        for (const uint8_t *d = img + dimp_rva; rd32(d + 4) != 0; d += 32) {
            uint64_t *iat = (uint64_t *)(img + rd32(d + 12));
            for (int k = 0; iat[k] != 0; ++k) { iat[k] = make_stub((const char *)(img + rd32(d + 4)), "(delay import)"); ++trapped; }
        }
    if (n_bound) *n_bound = bound;
    if (n_answered) *n_answered = answered;
    return trapped;
}

/* signal handlers and the Windows TEB the code expects behind GS: MSVC's __chkstk reads the stack limit at gs:[0x10] (0 = "already
 * committed", which on Linux it is); the CRT start-up reads the TEB's self pointer at gs:[0x30] and the stack base at gs:[0x08];
 * __declspec(thread) data goes through gs:[0x58] (an array of per-module blocks: zero pages here). */
int pe_prepare(void)
{
    g_no_sse = getenv("PE_RUN_NO_SSE") != NULL;
    g_verbose = getenv("PE_RUN_VERBOSE") != NULL;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGILL, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
    sigaction(SIGFPE, &sa, NULL);
    uint8_t *teb = (uint8_t *)mmap(NULL, 4 * 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (teb == MAP_FAILED) return -1;
    uint64_t self = (uint64_t)(uintptr_t)teb, stack_base = (uint64_t)-4096, tls_array = self + 4096;
    memcpy(teb + 0x30, &self, 8);
    memcpy(teb + 0x08, &stack_base, 8);
    memcpy(teb + 0x58, &tls_array, 8);
    for (int i = 0; i < 8; ++i) { uint64_t blk = self + 2 * 4096; memcpy(teb + 4096 + 8 * i, &blk, 8); }
    if (syscall(SYS_arch_prctl, 0x1001 /* ARCH_SET_GS */, teb) != 0) return -1;
    return 0;
}

/* ---- confinement ------------------------------------------------------------------------------------------------------------ */
#define ALLOW(nr) BPF_JUMP(BPF_JMP | BPF_JEQ | BPF_K, (nr), 0, 1), BPF_STMT(BPF_RET | BPF_K, SECCOMP_RET_ALLOW)

/* Drops privileges, moves to `scratch` (an empty directory the parent made) and installs the syscall filter on every thread.
 * Returns 0 when the process is confined; anything else means it is NOT and pe_call will refuse. */
int pe_lockdown(const char *scratch)
{
    if (g_locked) return 0;
    if (scratch && chdir(scratch) != 0) return -2;
    if (geteuid() == 0) {
        if (setgroups(0, NULL) != 0 || setresgid(65534, 65534, 65534) != 0 || setresuid(65534, 65534, 65534) != 0) return -3;
        if (setuid(0) == 0) return -4;                                                          /* must not be able to come back */
    }
    if (prctl(PR_SET_NO_NEW_PRIVS, 1, 0, 0, 0) != 0) return -5;
    static struct sock_filter f[] = {
        BPF_STMT(BPF_LD | BPF_W | BPF_ABS, offsetof(struct seccomp_data, arch)),
        BPF_JUMP(BPF_JMP | BPF_JEQ | BPF_K, AUDIT_ARCH_X86_64, 1, 0),
        BPF_STMT(BPF_RET | BPF_K, SECCOMP_RET_KILL_PROCESS),
        BPF_STMT(BPF_LD | BPF_W | BPF_ABS, offsetof(struct seccomp_data, nr)),
        BPF_JUMP(BPF_JMP | BPF_JGE | BPF_K, 0x40000000u, 0, 1),                                 /* the x32 table: no */
        BPF_STMT(BPF_RET | BPF_K, SECCOMP_RET_KILL_PROCESS),
        /* compute, answer, leave: descriptors already open, memory, signals of its own, clocks */
        ALLOW(SYS_read), ALLOW(SYS_write), ALLOW(SYS_readv), ALLOW(SYS_writev), ALLOW(SYS_pread64), ALLOW(SYS_pwrite64),
        ALLOW(SYS_close), ALLOW(SYS_fstat), ALLOW(SYS_lseek),
        ALLOW(SYS_mmap), ALLOW(SYS_munmap), ALLOW(SYS_mremap), ALLOW(SYS_mprotect), ALLOW(SYS_brk), ALLOW(SYS_madvise),
        ALLOW(SYS_futex), ALLOW(SYS_rt_sigaction), ALLOW(SYS_rt_sigprocmask), ALLOW(SYS_rt_sigreturn), ALLOW(SYS_sigaltstack),
        ALLOW(SYS_clock_gettime), ALLOW(SYS_gettimeofday), ALLOW(SYS_getrandom), ALLOW(SYS_sched_yield),
        ALLOW(SYS_getpid), ALLOW(SYS_gettid), ALLOW(SYS_getuid), ALLOW(SYS_geteuid), ALLOW(SYS_exit), ALLOW(SYS_exit_group),
        BPF_STMT(BPF_RET | BPF_K, SECCOMP_RET_ERRNO | EPERM),                                   /* everything else: refused */
    };
    struct sock_fprog prog = {(unsigned short)(sizeof f / sizeof f[0]), f};
    if (syscall(SYS_seccomp, SECCOMP_SET_MODE_FILTER, SECCOMP_FILTER_FLAG_TSYNC, &prog) != 0) return -6;
    /* proof, not hope: an open and a socket must now fail */
    if (syscall(SYS_openat, -100, "/proc/self/status", 0) >= 0 || errno != EPERM) return -7;
    if (syscall(SYS_socket, 2, 1, 0) >= 0 || errno != EPERM) return -8;
    g_locked = 1;
    return 0;
}

int pe_locked(void) { return g_locked; }

typedef uint64_t (MS *ms_vfn)(uint64_t, ...);

/* Calls the function at `addr` (Microsoft x64 convention) with eight 8-byte argument slots.  Slots 1..3 are handed over as a variadic
 * call hands floating-point arguments -- the same bits in the integer register AND in xmm1..3 -- so one shape serves pointer, integer,
 * float and double parameters alike (a float's bits in the low half of its slot); slot 0 is integer-only (every function called here
 * takes a pointer first); slots 4..7 go on the stack, where the type does not matter.  A callee that takes fewer ignores the rest.
 * Returns rax.  Refuses (exit 76) in a process that is not confined. */
uint64_t pe_call(uint64_t addr, const uint64_t *a)
{
    if (!g_locked) { say("pe_run: pe_call before pe_lockdown -- refusing to run reference code unconfined\n"); _exit(76); }
    if (g_verbose) { char w[160], m[220]; snprintf(m, sizeof m, "pe_run: call %s\n", where(addr, w, sizeof w)); say(m); }
    double d1, d2, d3;
    memcpy(&d1, &a[1], 8); memcpy(&d2, &a[2], 8); memcpy(&d3, &a[3], 8);
    return ((ms_vfn)(uintptr_t)addr)(a[0], d1, d2, d3, a[4], a[5], a[6], a[7]);
}
