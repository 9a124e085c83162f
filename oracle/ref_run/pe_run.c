/*
 * pe_run.c -- TEST INFRASTRUCTURE (oracle side): run import-free leaf functions of the reference's SHIPPED x64 binaries in place.
 *
 * The reference cannot be built in this image, but it ships the x86-64 machine code it runs
 * (/root/reference/stitching/OpenCV2.4.11-Stitching-64bit/x64/Release/opencv_{core,imgproc,stitching}2411.dll).  This maps one of those
 * PE32+ images read-only-from-disk into memory at its preferred base (no relocation needed), points EVERY import at a trap (so the code
 * can reach no library, no allocator, no system call -- only its own arithmetic), and calls a function at a given address with the
 * Microsoft x64 calling convention.  Nothing of the reference is copied or rebuilt: no headers, no libraries -- a function that needs an
 * import on the path taken aborts the process (exit code 77) and is simply not usable this way.  Four imports are ANSWERED instead of
 * trapped, and listed where they are defined below: the reference's own cv::checkHardwareSupport, and the C runtime's memset / memcpy /
 * memmove.
 *
 * tests/ref_run.py drives it (a subprocess per DLL: all three images want the same base) with hand-laid cv::Mat headers over numpy
 * buffers, compares what the reference's own code returns with oracle/*.c, and writes the golden vectors under tests/golden/ref_*.npz.
 * Runs only where /root/reference exists.
 *
 * Build: gcc -O1 -shared -fPIC -o libpe_run.so pe_run.c
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#ifndef MAP_FIXED_NOREPLACE
#define MAP_FIXED_NOREPLACE 0x100000
#endif

static void trap(void)
{
    char msg[160];
    int n = snprintf(msg, sizeof msg, "pe_run: the reference code called an IMPORT (from %p) on this path -- not a leaf; aborting\n",
                     __builtin_return_address(0));
    ssize_t r = write(2, msg, (size_t)n);
    (void)r;
    _exit(77);
}

/* The ONE import that is answered instead of trapped: cv::checkHardwareSupport(int feature) of the reference's own opencv_core2411.dll,
 * which the imgproc loops ask before taking their SSE bodies.  The real function returns what cpuid said when the DLL was loaded; here
 * it answers as every x86-64 CPU does -- MMX (1), SSE (2), SSE2 (3): yes; anything newer: no -- so that the SSE2 bodies the audit read
 * (and the reference ran) are the ones that execute.  PE_RUN_NO_SSE=1 answers "no" to everything instead (the scalar loops alone). */
static int g_no_sse;
static uint64_t __attribute__((ms_abi)) check_hardware_support(int feature)
{
    return (!g_no_sse && feature >= 1 && feature <= 3) ? 1u : 0u;
}

/* a fault inside the mapped code: say where (instruction and address), then exit 78 */
#include <signal.h>
#include <ucontext.h>
static void on_fault(int sig, siginfo_t *si, void *uc_)
{
    ucontext_t *uc = (ucontext_t *)uc_;
    char msg[200];
    int n = snprintf(msg, sizeof msg, "pe_run: signal %d at rip %p touching %p (rsp %p)\n", sig, (void *)uc->uc_mcontext.gregs[REG_RIP], si->si_addr,
                     (void *)uc->uc_mcontext.gregs[REG_RSP]);
    ssize_t r = write(2, msg, (size_t)n);
    (void)r;
    _exit(78);
}

/* ... and the C runtime's three byte movers, answered by the host C library's: memset, memcpy, memmove (MSVCR100 is not in the tree).  They
 * move or fill bytes -- no arithmetic, nothing a restatement could get wrong -- and without them a loop that zero-fills a local array or
 * copies a border pixel could not run.  Every other import of the C runtime (allocation, math, I/O, locale ...) stays trapped. */
static void *__attribute__((ms_abi)) crt_memset(void *d, int c, size_t n) { return memset(d, c, n); }
static void *__attribute__((ms_abi)) crt_memcpy(void *d, const void *s, size_t n) { return memcpy(d, s, n); }
static void *__attribute__((ms_abi)) crt_memmove(void *d, const void *s, size_t n) { return memmove(d, s, n); }

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* Maps the image; returns its base (0 on failure).  *n_imports = IAT slots pointed at the trap. */
uint64_t pe_load(const char *path, int *n_imports)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    g_no_sse = getenv("PE_RUN_NO_SSE") != NULL;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGILL, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
    sigaction(SIGFPE, &sa, NULL);
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *file = (uint8_t *)malloc((size_t)len);
    if (!file || fread(file, 1, (size_t)len, f) != (size_t)len) { fclose(f); free(file); return 0; }
    fclose(f);
    if (len < 0x200 || rd16(file) != 0x5A4D) { free(file); return 0; }
    const uint8_t *nt = file + rd32(file + 0x3C);
    if (rd32(nt) != 0x00004550 || rd16(nt + 4) != 0x8664) { free(file); return 0; }           /* "PE\0\0", AMD64 */
    const int n_sections = rd16(nt + 6);
    const int opt_size = rd16(nt + 20);
    const uint8_t *opt = nt + 24;
    if (rd16(opt) != 0x20B) { free(file); return 0; }                                           /* PE32+ */
    const uint64_t base = rd64(opt + 24);
    const uint32_t image_size = rd32(opt + 56), headers_size = rd32(opt + 60);
    uint8_t *img = (uint8_t *)mmap((void *)base, image_size, PROT_READ | PROT_WRITE | PROT_EXEC,
                                   MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED_NOREPLACE, -1, 0);
    if (img == MAP_FAILED || (uint64_t)img != base) { free(file); return 0; }
    memcpy(img, file, headers_size);
    const uint8_t *sec = opt + opt_size;
    for (int i = 0; i < n_sections; ++i, sec += 40) {
        const uint32_t vsize = rd32(sec + 8), va = rd32(sec + 12), raw = rd32(sec + 16), off = rd32(sec + 20);
        const uint32_t n = raw < vsize ? raw : vsize;
        if ((uint64_t)va + n <= image_size && (long)off + n <= len) memcpy(img + va, file + off, n);
    }
    /* every import address table slot -> the trap */
    int slots = 0;
    const uint32_t imp_rva = rd32(opt + 112 + 8 * 1), imp_size = rd32(opt + 112 + 8 * 1 + 4);
    if (imp_rva && imp_size)
        for (const uint8_t *d = img + imp_rva; rd32(d + 12) != 0; d += 20) {                    /* IMAGE_IMPORT_DESCRIPTOR, Name != 0 */
            uint64_t *iat = (uint64_t *)(img + rd32(d + 16));
            const uint64_t *names = (const uint64_t *)(img + (rd32(d) ? rd32(d) : rd32(d + 16)));
            for (int k = 0; names[k] != 0; ++k) {
                iat[k] = (uint64_t)(uintptr_t)&trap;
                ++slots;
                if (!(names[k] >> 63) && names[k] + 2 < image_size &&
                    strcmp((const char *)(img + names[k] + 2), "?checkHardwareSupport@cv@@YA_NH@Z") == 0)
                    iat[k] = (uint64_t)(uintptr_t)&check_hardware_support;
                if (!(names[k] >> 63) && names[k] + 2 < image_size) {
                    const char *nm = (const char *)(img + names[k] + 2);
                    if (strcmp(nm, "memset") == 0) iat[k] = (uint64_t)(uintptr_t)&crt_memset;
                    else if (strcmp(nm, "memcpy") == 0) iat[k] = (uint64_t)(uintptr_t)&crt_memcpy;
                    else if (strcmp(nm, "memmove") == 0) iat[k] = (uint64_t)(uintptr_t)&crt_memmove;
                }
            }
        }
    const uint32_t dimp_rva = rd32(opt + 112 + 8 * 13);                                        /* delay imports: their IAT too */
    if (dimp_rva)
        for (const uint8_t *d = img + dimp_rva; rd32(d + 4) != 0; d += 32) {
            uint64_t *iat = (uint64_t *)(img + rd32(d + 12));
            for (int k = 0; iat[k] != 0; ++k) { iat[k] = (uint64_t)(uintptr_t)&trap; ++slots; }
        }
    free(file);
    if (n_imports) *n_imports = slots;
    /* MSVC's __chkstk reads the thread's stack limit at gs:[0x10] (the Windows TEB) before a frame of more than a page; Linux leaves GS
     * free, so give it a page of zeros: a limit of 0 means "the stack is already committed", which on Linux it is. */
    void *teb = mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (teb == MAP_FAILED || syscall(SYS_arch_prctl, 0x1001 /* ARCH_SET_GS */, teb) != 0) return 0;
    return base;
}

typedef uint64_t (__attribute__((ms_abi)) *ms_fn8)(uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);

/* Calls the function at `addr` with up to eight integer / pointer arguments (Microsoft x64 convention; a callee that takes fewer
 * ignores the rest).  Returns rax. */
uint64_t pe_call(uint64_t addr, const uint64_t *a)
{
    return ((ms_fn8)(uintptr_t)addr)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
}

typedef double (__attribute__((ms_abi)) *ms_fn_d)(double, double, double, double);
double pe_call_d(uint64_t addr, const double *a)
{
    return ((ms_fn_d)(uintptr_t)addr)(a[0], a[1], a[2], a[3]);
}
