#!/usr/bin/perl
# TEST INFRASTRUCTURE — token-level GLSL 3.30 -> C++ rewrite for oracle/glsl_on_cpu/glsl.h.  Reads one shader of the
# reference where it lies, writes C++ to stdout (a pipe into g++: nothing from the reference is stored).
#   usage: glsl2cpp.pl <namespace> <shader file>
# Rewrites: #include expansion; #version / layout(...) / in / out / uniform / flat qualifiers dropped (interface
# variables become namespace-scope variables the bridge reads and writes by name); unsized geometry inputs become
# one-element arrays; unsuffixed floating literals get an f (GLSL literals are float, C++ ones double); multi-component
# swizzles become member calls; `discard` sets a flag and returns; main -> shader_main.
use strict;
use File::Basename;
my ($ns, $path) = @ARGV;
sub slurp {
  my ($p) = @_;
  open(my $f, '<', $p) or die "cannot read $p";
  local $/;
  my $s = <$f>;
  my $dir = dirname($p);
  $s =~ s{^\s*#include\s+"([^"]+)"\s*$}{slurp("$dir/$1")}gme;
  return $s;
}
my $s = slurp($path);
$s =~ s{/\*.*?\*/}{}gs;
$s =~ s{//[^\n]*}{}g;
$s =~ s{^\s*#version[^\n]*$}{}gm;
$s =~ s{^\s*layout\s*\([^)]*\)\s*(in|out)\s*;\s*$}{}gm;
$s =~ s{layout\s*\([^)]*\)}{}g;
$s =~ s{^\s*(?:flat\s+)?(?:in|out|uniform)\s+(?=\w)}{}gm;
$s =~ s{\[\s*\]\s*;}{[1];}g;
$s =~ s{(?<![\w.])(\d+\.\d*|\.\d+)([eE][-+]?\d+)?(?![\w.])}{$1 . ($2 // '') . 'f'}ge;
$s =~ s{\b(\w+)\.xyz\s*=(?!=)\s*([^;]+);}{$1.set_xyz($2);}g;
$s =~ s{\.(xyz|xy|zw)\b(?!\s*\()}{.$1()}g;
$s =~ s{\bdiscard\s*;}{\{ discard_flag = true; return; \}}g;
$s =~ s{\bvoid\s+main\s*\(\s*\)}{void shader_main()}g;
print "namespace glsl { namespace $ns {\n$s\n} }\n";
