/* TEST INFRASTRUCTURE ONLY.  An LD_PRELOAD interposer that opens ASM_REDIRECT_TO whenever a program opens the path
 * ASM_REDIRECT_FROM.  The reference's GASMA/benchmark/benchmark.cpp:28 reads a hard-coded "/home/zhenhao/..." file; the drop-in
 * test runs that program UNMODIFIED, so the file name it asks for is answered at the C-library boundary instead (a test may not
 * be able to create /home/zhenhao on the machine it runs on).  Built by `make -C oracle shim` into oracle/_ref/. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char* redirect(const char* path) {
    const char* from = getenv("ASM_REDIRECT_FROM");
    const char* to = getenv("ASM_REDIRECT_TO");
    if (path && from && to && strcmp(path, from) == 0) return to;
    return path;
}

#define REAL(name) ((real_##name) ? (real_##name) : ((real_##name) = dlsym(RTLD_NEXT, #name)))

FILE* fopen(const char* path, const char* mode) {
    static FILE* (*real_fopen)(const char*, const char*);
    return REAL(fopen)(redirect(path), mode);
}
FILE* fopen64(const char* path, const char* mode) {
    static FILE* (*real_fopen64)(const char*, const char*);
    return REAL(fopen64)(redirect(path), mode);
}
int open(const char* path, int flags, ...) {
    static int (*real_open)(const char*, int, ...);
    mode_t mode = 0;
    if (flags & (O_CREAT | O_TMPFILE)) {
        va_list ap;
        va_start(ap, flags);
        mode = (mode_t)va_arg(ap, int);
        va_end(ap);
    }
    return REAL(open)(redirect(path), flags, mode);
}
int open64(const char* path, int flags, ...) {
    static int (*real_open64)(const char*, int, ...);
    mode_t mode = 0;
    if (flags & (O_CREAT | O_TMPFILE)) {
        va_list ap;
        va_start(ap, flags);
        mode = (mode_t)va_arg(ap, int);
        va_end(ap);
    }
    return REAL(open64)(redirect(path), flags, mode);
}
int openat(int dirfd, const char* path, int flags, ...) {
    static int (*real_openat)(int, const char*, int, ...);
    mode_t mode = 0;
    if (flags & (O_CREAT | O_TMPFILE)) {
        va_list ap;
        va_start(ap, flags);
        mode = (mode_t)va_arg(ap, int);
        va_end(ap);
    }
    return REAL(openat)(dirfd, redirect(path), flags, mode);
}
